#!/bin/bash
# One more pair of counter passes on the headline workload: the L2 -> fabric read requests by SIZE (the exact byte
# count behind FETCH_SIZE's x2 correction) and by TARGET (DRAM address space / GMI / IO).  rocprofv3 on gfx950 lists
# no Infinity-Cache (MALL) hit / miss counter (`rocprofv3 -L`: none), so HBM bytes stay a bracket.
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r04_prof
RAW=/tmp/prof_raw_ea
rm -rf $RAW; mkdir -p $OUT $RAW
cd /tmp && export TMPDIR=/tmp
LEAN="--cpu-baseline off --host-boundary off --edge-cut-probe off --small-batches off --other-configs= --verify-oracle off"
B="python $R/bench.py --workload c3 $LEAN --steps 5 --warmup 1 --detail-out $OUT/ea_detail.json"
i=0
for pass in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" "TCC_EA0_RDREQ_DRAM_32B_sum TCC_EA0_RDREQ_GMI_32B_sum TCC_EA0_RDREQ_IO_32B_sum TCC_EA0_WRREQ_64B_sum"; do
  i=$((i+1))
  timeout 900 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $RAW/p$i -o p -- $B > $OUT/ea_bench_$i.json 2> $OUT/ea_$i.err
  for f in $(find $RAW/p$i -name '*counter_collection.csv'); do (head -1 $f; grep glx_aggregate $f) > $OUT/ea_pass${i}_counter_collection.csv; done
done
tail -n 2 $OUT/ea_*.err | head; ls -la $OUT | grep ea_
