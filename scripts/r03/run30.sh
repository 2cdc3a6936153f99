#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03_run30; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -q -x > $O/pytest_all.log 2>&1; tail -6 $O/pytest_all.log
bash scripts/r03/profile.sh 2>&1 | tail -30
