#!/bin/bash
# Round-5 profiles: rocprofv3 kernel traces + PMC passes per workload, and the before / after pair of the
# aggregation kernel (GLX_AGG_LEGACY=1 = the round-3 kernel) on the headline workload.
#   gpurun --timeout 3000 -- bash scripts/r05/profile.sh [c3 c2 c5 c4]
# Raw outputs stay in /tmp; per-kernel CSV rows of the glx kernels go to gpurun_out/r05_prof/; scripts/r05/summarize.py
# turns them into profiles/r05/.  Counters are collected in runs of their own with --kernel-trace only (never with
# --sys-trace / --hip-trace: gpurun refuses that combination).
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r05_prof
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
  if [ -z "$ONLY_PASS" ]; then  # ONLY_PASS=WRITE_SIZE: redo one counter pass (rocprofv3's own crash on c4 is intermittent)
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $RAW/${wl}_trace -o t -- $B --steps 20 --warmup 5 --roofline-probes off > $OUT/${wl}_bench_trace.json 2> $OUT/${wl}_trace.err
  keep $RAW/${wl}_trace $OUT/${wl}_trace
  fi
  # c4: counter collection over the 1.6 B-edge build's dispatches crashes rocprofv3 (SIGSEGV in its own thread, r03 and
  # r05): collect counters for the path's kernels only
  ONLY=""; if [ $wl = c4 ]; then ONLY="--kernel-include-regex glx_aggregate|glx_sample|glx_rwor"; fi
  for pass in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
    tag=$(echo $pass | cut -d' ' -f1)
    if [ -n "$ONLY_PASS" ] && [ "$ONLY_PASS" != "$tag" ]; then continue; fi
    timeout 900 rocprofv3 --pmc $pass $ONLY --kernel-trace --output-format csv -d $RAW/${wl}_$tag -o p -- $B --steps 5 --warmup 1 > $OUT/${wl}_bench_$tag.json 2> $OUT/${wl}_$tag.err
    keep $RAW/${wl}_$tag $OUT/${wl}_$tag
    rm -rf $RAW/${wl}_$tag
  done
  rm -rf $RAW/${wl}_trace
done
# the headline command as the driver runs it (three request-shape legs in the timed region), kernel trace only: shows the
# segment bookkeeping kernels of the explicit-segment_ids leg beside the reduce
if echo "$WLS" | grep -q c3; then
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $RAW/c3_shapes -o t -- python $R/bench.py --workload c3 --cpu-baseline off --host-boundary off --edge-cut-probe off --small-batches off --other-configs= --roofline-probes off --steps 20 --warmup 3 --detail-out $OUT/c3_shapes_detail.json > $OUT/c3_shapes_bench_trace.json 2> $OUT/c3_shapes_trace.err
  keep $RAW/c3_shapes $OUT/c3_shapes_trace
  rm -rf $RAW/c3_shapes
fi
# before / after on the headline workload: same command, round-3 kernel (GLX_AGG_LEGACY=1) vs the grouped kernel
if echo "$WLS" | grep -q c3 && [ -n "$WITH_AB" ]; then  # SKIP_AB=1: traces and traffic passes only
  B="python $R/bench.py --workload c3 $LEAN --roofline-probes off --steps 5 --warmup 1"
  for variant in legacy grouped; do
    if [ $variant = legacy ]; then export GLX_AGG_LEGACY=1; else unset GLX_AGG_LEGACY; fi
    for pass in "SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_VALU" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES" "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE"; do
      tag=$(echo $pass | cut -d' ' -f1)
      timeout 900 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $RAW/ab_${variant}_$tag -o p -- $B --detail-out $OUT/ab_${variant}_detail.json > $OUT/ab_${variant}_$tag.json 2> $OUT/ab_${variant}_$tag.err
      keep $RAW/ab_${variant}_$tag $OUT/ab_${variant}_$tag
      rm -rf $RAW/ab_${variant}_$tag
    done
  done
  unset GLX_AGG_LEGACY
fi
tail -n 2 $OUT/*.err | grep -v "^$" | head -40
ls $OUT | wc -l; du -sh $OUT
