#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03_run29; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_dist_store.py tests/test_gpu_dist_ledger.py tests/test_gpu_sharded.py -m gpu -q -x > $O/pytest.log 2>&1; grep -E "passed|failed|error" $O/pytest.log | tail -3
for rep in 1 2; do
for m in queue noqueue; do
if [ $m = noqueue ]; then export GLX_RESOLVE_NO_QUEUE=1; else unset GLX_RESOLVE_NO_QUEUE; fi
echo -n "$m: "; GRAPH_REPLICA=1 LEDGER=1 MERGED=1 timeout 600 python scripts/edge_cut_p8_probe.py 8 0.25 10 2>&1 | grep -E "^P = " | cut -c48-130
done; done
unset GLX_RESOLVE_NO_QUEUE
bash scripts/r03/run28.sh
