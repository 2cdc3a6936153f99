#!/bin/bash
# round 3, GPU call 2: sub-graph + count/stats getters
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03_run2; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_round3.py tests/test_host_cpp.py tests/test_gpu_pyapi.py -m gpu -q -x > $O/pytest.log 2>&1; tail -25 $O/pytest.log
