#!/bin/bash
# Round 2, GPU run 26: stream priority of the short-kernel stages, world-1 edge-cut bench A/B.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD
export TMPDIR=/tmp
O=$R/gpurun_out/r02_run26
mkdir -p $O
for M in high normal high normal; do
  GLX_DIST_NO_SHORTCUT=1 timeout 600 python bench.py --gpus 1 --force-sharded --steps 20 --warmup 5 --cpu-baseline off --stage-priority $M > $O/bench_w1_$M.json 2> $O/bench_w1_$M.log
  python -c "import json; r=json.load(open('$O/bench_w1_$M.json')); print('$M', r['placements'])"
done
RAW=/tmp/prof_w1; rm -rf $RAW; mkdir -p $RAW
(cd /tmp && GLX_DIST_NO_SHORTCUT=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $RAW -o w1 -- python $R/bench.py --gpus 1 --force-sharded --steps 20 --warmup 5 --cpu-baseline off --features sharded > /dev/null 2>&1)
for f in $(find $RAW -name '*kernel_stats.csv'); do grep "glx_" $f | cut -c1-200 > $O/kernel_stats_w1.csv; done
grep "resolve\|assign\|finalize\|stitch2\|part_\|lookup" $O/kernel_stats_w1.csv | cut -c1-160
