#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03_run12; mkdir -p $O
cd $R
SECONDS=0
timeout 1200 python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err; echo "rc=$? wall=$SECONDS"
grep "other config\|cpu baseline" $O/bench_default.err | cut -c1-3000
