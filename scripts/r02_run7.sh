#!/bin/bash
# Round 2, GPU run 7: halo-set sizing fix: edge-cut path at P = 8 / 2 on one GPU again, dist tests.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD
export TMPDIR=/tmp
O=$R/gpurun_out/r02_run7
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_dist_store.py tests/test_gpu_two_ranks.py tests/test_gpu_sharded.py tests/test_host_cpp.py -x -q -m gpu > $O/pytest_dist.log 2>&1
echo "pytest_dist rc=$?" | tee -a $O/status.txt
tail -5 $O/pytest_dist.log
for cfg in "8 0.10" "8 0.25" "8 0.0" "2 0.10" "4 0.10"; do
  set -- $cfg
  timeout 900 python scripts/edge_cut_p8_probe.py $1 $2 6 > $O/edge_cut_p$1_hot$2.txt 2>&1
  echo "p$1 hot$2 rc=$?" | tee -a $O/status.txt
  tail -3 $O/edge_cut_p$1_hot$2.txt
done
RAW=/tmp/prof_p8; rm -rf $RAW; mkdir -p $RAW
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $RAW -o p8 -- python $R/scripts/edge_cut_p8_probe.py 8 0.10 6 > $O/edge_cut_p8_trace.txt 2>&1)
for f in $(find $RAW -name '*kernel_stats.csv'); do (head -1 $f; grep "glx_\|rocclr" $f) > $O/edge_cut_p8_kernel_stats.csv; done
head -16 $O/edge_cut_p8_kernel_stats.csv
GLX_DIST_NO_SHORTCUT=1 timeout 600 python bench.py --gpus 1 --force-sharded --steps 20 --warmup 5 --cpu-baseline off --verify > $O/bench_w1_sharded.json 2> $O/bench_w1_sharded.log
python -c "import json; r=json.load(open('$O/bench_w1_sharded.json')); print(r['placements'], r['verified_sharded_equals_unpartitioned'])"
