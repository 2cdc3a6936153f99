#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=$PWD/gpurun_out/r02_run40
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_two_ranks.py -x -q -m gpu --timeout 600 > $O/pytest.log 2>&1
echo "rc=$?"
tail -15 $O/pytest.log | cut -c1-300
