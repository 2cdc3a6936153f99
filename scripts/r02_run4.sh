#!/bin/bash
# Round 2, GPU run 4: wave alias build (load-time + filtered), pinned response blocks, two-stream graph steps.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD
export TMPDIR=/tmp
O=$R/gpurun_out/r02_run4
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_filter.py tests/test_gpu_fuzz.py tests/test_gpu_configs.py tests/test_host_cpp.py -x -q -m gpu --timeout 600 > $O/pytest_new.log 2>&1
echo "pytest_new rc=$?" | tee -a $O/status.txt
tail -15 $O/pytest_new.log
timeout 1500 python -m pytest tests -q -m gpu --timeout 900 > $O/pytest_all.log 2>&1
echo "pytest_all rc=$?" | tee -a $O/status.txt
tail -8 $O/pytest_all.log
timeout 600 python scripts/filter_bench.py --index > $O/filter_bench_index.txt 2>&1
timeout 600 python scripts/filter_bench.py > $O/filter_bench_scan.txt 2>&1
echo "filter bench rc=$?" | tee -a $O/status.txt
cat $O/filter_bench_index.txt
for B in 1024 8192; do
  for GS in 1 2 3; do
    timeout 300 python bench.py --batch $B --steps 400 --warmup 40 --cpu-baseline off --roofline-probes off --graph on --graph-streams $GS > $O/bench_b${B}_graph_s$GS.json 2> $O/bench_b${B}_graph_s$GS.log
    python -c "import json; r=json.load(open('$O/bench_b${B}_graph_s$GS.json')); print('B0=$B graph streams=$GS', r['ms_per_step'], r['value'])"
  done
done
timeout 300 python bench.py --steps 20 --warmup 5 --cpu-baseline off > $O/bench_n1.json 2> $O/bench_n1.log
python -c "import json; r=json.load(open('$O/bench_n1.json')); print('B0=65536', r['ms_per_step'], r['value'])"
grep -i "built in\|generated" $O/bench_n1.log
H=./graph-learn_amd/lib/host_path_bench
for T in 1 8 32; do
  timeout 300 $H $T 1024 20 >> $O/host_path_pinned.txt 2>&1
  GLX_HOST_PINNED_RESPONSES=0 timeout 300 $H $T 1024 20 >> $O/host_path_pageable.txt 2>&1
done
echo pinned; grep threads $O/host_path_pinned.txt; echo pageable; grep threads $O/host_path_pageable.txt
