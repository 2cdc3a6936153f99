#!/bin/bash
# Round-6 profiles: per workload ONE rocprofv3 --kernel-trace --stats run of the lean bench command WITH the roofline
# probes on (so the cache-free launches `roofline.frac` is taken from are in the trace: VERDICT r05 next-3), then the
# PMC passes FETCH_SIZE / WRITE_SIZE / TCC, each in a run of its own with --kernel-trace only.
#   gpurun --timeout 3000 -- bash scripts/r06/profile.sh [c3 c2 c4 c5]        PMC=0: traces only
# Raw outputs stay in /tmp; the glx rows of every CSV go to gpurun_out/r06_prof/; scripts/r06/summarize.py turns them into
# profiles/r06/.
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r06_prof
RAW=/tmp/prof_raw
rm -rf $RAW; mkdir -p $OUT $RAW
WLS=${@:-c3}
cd /tmp && export TMPDIR=/tmp
LEAN="--cpu-baseline off --host-boundary off --edge-cut-probe off --small-batches off --other-configs= --verify-oracle off --request-shape-legs off"
keep() {  # keep <raw dir> <out prefix>: the glx rows of every trace / counter CSV
  for f in $(find $1 -name '*kernel_trace.csv' -o -name '*counter_collection.csv' -o -name '*kernel_stats.csv'); do
    b=$(basename $f)
    (head -1 $f; grep glx_ $f) > $2_$b
  done
}
for wl in $WLS; do
  B="python $R/bench.py --workload $wl $LEAN --detail-out $OUT/${wl}_detail.json"
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $RAW/${wl}_trace -o t -- $B --steps 20 --warmup 5 > $OUT/${wl}_bench_trace.json 2> $OUT/${wl}_trace.err
  keep $RAW/${wl}_trace $OUT/${wl}_trace
  cp $OUT/${wl}_detail.json $OUT/${wl}_detail_trace.json 2>/dev/null
  rm -rf $RAW/${wl}_trace
  if [ "${PMC:-1}" = "1" ]; then
    ONLY=""; if [ $wl = c4 ]; then ONLY="--kernel-include-regex glx_aggregate|glx_sample|glx_rwor"; fi
    for pass in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
      tag=$(echo $pass | cut -d' ' -f1)
      timeout 900 rocprofv3 --pmc $pass $ONLY --kernel-trace --output-format csv -d $RAW/${wl}_$tag -o p -- $B --steps 5 --warmup 1 > $OUT/${wl}_bench_$tag.json 2> $OUT/${wl}_$tag.err
      keep $RAW/${wl}_$tag $OUT/${wl}_$tag
      rm -rf $RAW/${wl}_$tag
    done
  fi
done
tail -n 2 $OUT/*.err | grep -v "^$" | head -40
ls $OUT | wc -l; du -sh $OUT
