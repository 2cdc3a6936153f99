#!/bin/bash
# Round 2, GPU run 3: id-filter fast path, request plans (hipGraph), partition probe, host boundary baseline.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD
export TMPDIR=/tmp
O=$R/gpurun_out/r02_run3
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_filter.py tests/test_gpu_fuzz.py tests/test_gpu_parity.py tests/test_host_cpp.py -x -q -m gpu --timeout 600 > $O/pytest_new.log 2>&1
echo "pytest_new rc=$?" | tee -a $O/status.txt
tail -15 $O/pytest_new.log
timeout 1500 python -m pytest tests -q -m gpu --timeout 900 > $O/pytest_all.log 2>&1
echo "pytest_all rc=$?" | tee -a $O/status.txt
tail -8 $O/pytest_all.log
timeout 600 python scripts/filter_bench.py > $O/filter_bench_scan.txt 2>&1
timeout 600 python scripts/filter_bench.py --index > $O/filter_bench_index.txt 2>&1
GLX_FILTER_NO_FAST_PATH=1 timeout 600 python scripts/filter_bench.py > $O/filter_bench_general.txt 2>&1
echo "filter bench rc=$?" | tee -a $O/status.txt
cat $O/filter_bench_scan.txt $O/filter_bench_index.txt
RAW=/tmp/prof_part; rm -rf $RAW; mkdir -p $RAW
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $RAW -o part -- python $R/scripts/partition_probe.py > $O/partition_probe.txt 2>&1)
for f in $(find $RAW -name '*kernel_stats.csv'); do (head -1 $f; grep glx_ $f) > $O/partition_kernel_stats.csv; done
for f in $(find $RAW -name '*kernel_trace.csv'); do (head -1 $f; grep glx_part_scan $f | head -60) > $O/partition_scan_trace.csv; done
cat $O/partition_probe.txt | tail -12
cat $O/partition_kernel_stats.csv
for B in 1024 8192; do
  for G in on off; do
    timeout 300 python bench.py --batch $B --steps 300 --warmup 30 --cpu-baseline off --roofline-probes off --graph $G > $O/bench_b${B}_graph_$G.json 2> $O/bench_b${B}_graph_$G.log
    python -c "import json; r=json.load(open('$O/bench_b${B}_graph_$G.json')); print('B0=$B graph=$G', r['ms_per_step'], r['value'])"
  done
done
timeout 300 ./graph-learn_amd/lib/host_path_bench > $O/host_path_bench.txt 2>&1
tail -12 $O/host_path_bench.txt
