#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03_run24; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_dist_store.py tests/test_gpu_dist_ledger.py tests/test_gpu_sharded.py -m gpu -q -x > $O/pytest.log 2>&1; grep -E "passed|failed|error" $O/pytest.log | tail -3
GRAPH_REPLICA=1 LEDGER=1 MERGED=1 timeout 600 python scripts/edge_cut_p8_probe.py 8 0.25 6 2>&1 | grep -E "^P = |count exchanges" | cut -c1-330
GRAPH_REPLICA=1 LEDGER=1 MERGED=1 GLX_RESOLVE_ONE_PASS=1 timeout 600 python scripts/edge_cut_p8_probe.py 8 0.25 6 2>&1 | grep -E "^P = " | cut -c1-330
