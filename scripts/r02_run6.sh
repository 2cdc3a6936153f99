#!/bin/bash
# Round 2, GPU run 6: edge-cut path at P = 8 on one GPU (full C3), admission control at the host boundary.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD
export TMPDIR=/tmp
O=$R/gpurun_out/r02_run6
mkdir -p $O
timeout 900 python scripts/edge_cut_p8_probe.py 8 0.10 6 > $O/edge_cut_p8_hot10.txt 2>&1
echo "p8 hot10 rc=$?" | tee -a $O/status.txt
tail -4 $O/edge_cut_p8_hot10.txt
timeout 900 python scripts/edge_cut_p8_probe.py 8 0.25 6 > $O/edge_cut_p8_hot25.txt 2>&1
echo "p8 hot25 rc=$?" | tee -a $O/status.txt
tail -4 $O/edge_cut_p8_hot25.txt
timeout 900 python scripts/edge_cut_p8_probe.py 2 0.10 6 > $O/edge_cut_p2_hot10.txt 2>&1
tail -4 $O/edge_cut_p2_hot10.txt
RAW=/tmp/prof_p8; rm -rf $RAW; mkdir -p $RAW
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $RAW -o p8 -- python $R/scripts/edge_cut_p8_probe.py 8 0.10 6 > $O/edge_cut_p8_trace.txt 2>&1)
for f in $(find $RAW -name '*kernel_stats.csv'); do (head -1 $f; grep "glx_\|rocclr" $f) > $O/edge_cut_p8_kernel_stats.csv; done
head -25 $O/edge_cut_p8_kernel_stats.csv
H=./graph-learn_amd/lib/host_path_bench
for T in 8 16 32 64; do
  timeout 300 $H $T 1024 20 >> $O/host_path_admission16.txt 2>&1
done
GLX_HOST_CALL_CONCURRENCY=8 timeout 300 $H 32 1024 20 >> $O/host_path_admission8.txt 2>&1
GLX_HOST_CALL_CONCURRENCY=12 timeout 300 $H 32 1024 20 >> $O/host_path_admission12.txt 2>&1
GLX_HOST_CALL_CONCURRENCY=0 timeout 300 $H 32 1024 20 >> $O/host_path_admission0.txt 2>&1
grep -h threads $O/host_path_admission*.txt
timeout 600 python -m pytest tests/test_host_cpp.py tests/test_gpu_pyapi.py -x -q -m gpu > $O/pytest_host.log 2>&1
tail -3 $O/pytest_host.log
