#!/bin/bash
# Round 2, GPU run 22: graph replica through the C++ runner + the Python SPMD sampler; two-rank bench.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD
export TMPDIR=/tmp
O=$R/gpurun_out/r02_run22
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_dist_store.py tests/test_gpu_two_ranks.py tests/test_host_cpp.py tests/test_gpu_pyapi.py -q -m gpu --timeout 600 > $O/pytest.log 2>&1
echo "pytest rc=$?" | tee -a $O/status.txt
tail -5 $O/pytest.log
