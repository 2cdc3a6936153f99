#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03_run19; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_dist_store.py -m gpu -q -x -k "rccl" 2>&1 | tail -15
