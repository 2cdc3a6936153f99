#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03_run11; mkdir -p $O
cd $R
timeout 900 python bench.py --steps 20 --warmup 5 --cpu-baseline off --host-boundary off --edge-cut-probe off --small-batches off --other-configs c4 > $O/bench.json 2> $O/bench.err; echo "rc=$?"
grep "other config" $O/bench.err | cut -c1-2500
