#!/bin/bash
# Round 2, GPU run 16: id == value filters with hit runs read in place from the row index.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD
export TMPDIR=/tmp
O=$R/gpurun_out/r02_run16
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_filter.py tests/test_gpu_dist_store.py tests/test_host_cpp.py -q -m gpu --timeout 600 > $O/pytest_filter.log 2>&1
echo "pytest_filter rc=$?" | tee -a $O/status.txt
tail -4 $O/pytest_filter.log
timeout 600 python scripts/filter_bench.py --index > $O/filter_bench_index.txt 2>&1
grep '"op"' $O/filter_bench_index.txt
timeout 600 python scripts/filter_bench.py > $O/filter_bench_scan.txt 2>&1
grep '"op"' $O/filter_bench_scan.txt
