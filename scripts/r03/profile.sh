#!/bin/bash
# round 3 profiles: the driver's command (full line) + rocprofv3 kernel traces and PMC passes per workload.
# Raw outputs stay in /tmp; small per-kernel CSVs go to gpurun_out/r03_prof/.
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r03_prof
RAW=/tmp/prof_raw
rm -rf $RAW; mkdir -p $OUT $RAW
cd $R
timeout 900 python bench.py --steps 20 --warmup 5 > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "default bench rc=$?"
cd /tmp && export TMPDIR=/tmp
LEAN="--cpu-baseline off --host-boundary off --edge-cut-probe off --small-batches off --other-configs= --verify-oracle off"
for wl in c3 c2 c5 c4; do
  B="python $R/bench.py --workload $wl $LEAN"
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $RAW/${wl}_trace -o t -- $B --steps 20 --warmup 5 --roofline-probes off > $OUT/${wl}_bench_trace.json 2> $OUT/${wl}_trace.err
  if [ $wl != c4 ]; then  # (counter collection over the 1.6 B-edge build dumped core in rocprofv3 once: trace only)
  timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $RAW/${wl}_fetch -o f -- $B --steps 5 --warmup 1 > $OUT/${wl}_bench_fetch.json 2> $OUT/${wl}_fetch.err
  timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $RAW/${wl}_write -o w -- $B --steps 5 --warmup 1 > $OUT/${wl}_bench_write.json 2> $OUT/${wl}_write.err
  fi
  for f in $(find $RAW/${wl}_trace -name '*kernel_stats.csv'); do cp $f $OUT/${wl}_kernel_stats.csv; done
  for d in trace fetch write; do
    for f in $(find $RAW/${wl}_$d -name '*kernel_trace.csv' -o -name '*counter_collection.csv'); do
      b=$(basename $f)
      (head -1 $f; grep glx_ $f) > $OUT/${wl}_${d}_$b
    done
  done
done
tail -2 $OUT/*.err | head -60
ls -la $OUT
