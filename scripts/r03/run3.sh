#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03_run3; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_round3.py -m gpu -q -x -k "conditional" > $O/pytest.log 2>&1; tail -30 $O/pytest.log
