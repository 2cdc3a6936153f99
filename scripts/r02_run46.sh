#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=$PWD/gpurun_out/r02_run46
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu --timeout 800 -k "plan" > $O/pytest.log 2>&1
echo "rc=$?"
tail -25 $O/pytest.log | cut -c1-250
