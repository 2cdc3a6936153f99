#!/bin/bash
# Round 2, GPU run 24: rank-select membership of the feature replica; dist tests; world-1 edge-cut bench A/B.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD
export TMPDIR=/tmp
O=$R/gpurun_out/r02_run24
mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_dist_store.py tests/test_gpu_two_ranks.py tests/test_host_cpp.py -q -m gpu --timeout 900 > $O/pytest_dist.log 2>&1
echo "pytest_dist rc=$?" | tee -a $O/status.txt
grep -n "passed\|failed" $O/pytest_dist.log | tail -2
for M in bitmap hash; do
  if [ $M = hash ]; then export GLX_DIST_NO_BITMAP=1; fi
  GLX_DIST_NO_SHORTCUT=1 timeout 600 python bench.py --gpus 1 --force-sharded --steps 20 --warmup 5 --cpu-baseline off --verify > $O/bench_w1_$M.json 2> $O/bench_w1_$M.log
  python -c "import json; r=json.load(open('$O/bench_w1_$M.json')); print('$M', r['placements'], r['verified_sharded_equals_unpartitioned'])"
done
unset GLX_DIST_NO_BITMAP
RAW=/tmp/prof_w1; rm -rf $RAW; mkdir -p $RAW
(cd /tmp && GLX_DIST_NO_SHORTCUT=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $RAW -o w1 -- python $R/bench.py --gpus 1 --force-sharded --steps 20 --warmup 5 --cpu-baseline off --features sharded > /dev/null 2>&1)
for f in $(find $RAW -name '*kernel_stats.csv'); do grep "glx_" $f | cut -c1-200 > $O/kernel_stats_w1.csv; done
grep "resolve\|assign\|finalize\|stitch2\|part_\|lookup" $O/kernel_stats_w1.csv | cut -c1-160
