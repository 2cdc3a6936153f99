#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03_run13; mkdir -p $O
cd $R
for flags in "--host-boundary off --edge-cut-probe off" "--small-batches off --edge-cut-probe off" "--small-batches off --host-boundary off"; do
  timeout 1200 python bench.py --steps 20 --warmup 5 --cpu-baseline off $flags --other-configs c4 > $O/bench.json 2> $O/bench.err; echo "== $flags rc=$?"
  grep "other config\|before the other" $O/bench.err | cut -c1-300
done
