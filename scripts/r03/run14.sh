#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03_run14; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_pyapi.py tests/test_gpu_parity.py -m gpu -q -x > $O/pytest.log 2>&1; tail -25 $O/pytest.log
