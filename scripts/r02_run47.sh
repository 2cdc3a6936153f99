#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=$PWD/gpurun_out/r02_run47
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_dist_store.py -x -q -m gpu --timeout 600 -k "deepwalk" > $O/pytest.log 2>&1
echo "rc=$?"
tail -12 $O/pytest.log | cut -c1-250
timeout 300 ./graph-learn_amd/lib/partition_stitch_unittest 2>&1 | tail -4
