#!/bin/bash
# Round 2, GPU run 5: zero-copy responses into pinned host blocks, graph stream counts, PCIe probe.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD
export TMPDIR=/tmp
O=$R/gpurun_out/r02_run5
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_host_cpp.py tests/test_gpu_pyapi.py tests/test_gpu_filter.py -x -q -m gpu --timeout 600 > $O/pytest_new.log 2>&1
echo "pytest_new rc=$?" | tee -a $O/status.txt
tail -15 $O/pytest_new.log
timeout 1500 python -m pytest tests -q -m gpu --timeout 900 > $O/pytest_all.log 2>&1
echo "pytest_all rc=$?" | tee -a $O/status.txt
tail -8 $O/pytest_all.log
python scripts/pcie_probe.py 2>&1 | tail -3 | tee $O/pcie_probe.txt
H=./graph-learn_amd/lib/host_path_bench
for T in 1 4 8 16 32; do
  timeout 300 $H $T 1024 20 >> $O/host_path_zero_copy.txt 2>&1
  GLX_HOST_ZERO_COPY=0 timeout 300 $H $T 1024 20 >> $O/host_path_pinned_copy.txt 2>&1
done
GLX_HOST_PINNED_RESPONSES=0 timeout 300 $H 8 1024 20 >> $O/host_path_pageable.txt 2>&1
GLX_HOST_PINNED_RESPONSES=0 timeout 300 $H 32 1024 20 >> $O/host_path_pageable.txt 2>&1
echo zero-copy; grep threads $O/host_path_zero_copy.txt; echo pinned+copy; grep threads $O/host_path_pinned_copy.txt; echo pageable; grep threads $O/host_path_pageable.txt
for B in 1024 8192; do
  for GS in 3 4 6; do
    timeout 300 python bench.py --batch $B --steps 400 --warmup 40 --cpu-baseline off --roofline-probes off --host-boundary off --graph on --graph-streams $GS > $O/bench_b${B}_graph_s$GS.json 2> $O/bench_b${B}_graph_s$GS.log
    python -c "import json; r=json.load(open('$O/bench_b${B}_graph_s$GS.json')); print('B0=$B graph streams=$GS', r['ms_per_step'], r['value'])"
  done
done
