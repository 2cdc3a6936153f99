#!/bin/bash
# Round 2, GPU run 9: block-aggregated atomics in the halo kernels, global in-degrees, solo P=8 probe.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD
export TMPDIR=/tmp
O=$R/gpurun_out/r02_run9
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_dist_store.py tests/test_gpu_two_ranks.py tests/test_gpu_sharded.py tests/test_host_cpp.py tests/test_gpu_filter.py tests/test_gpu_pyapi.py -x -q -m gpu > $O/pytest_dist.log 2>&1
echo "pytest_dist rc=$?" | tee -a $O/status.txt
tail -5 $O/pytest_dist.log
for cfg in "8 0.10 solo" "8 0.25 solo" "8 0.0 solo" "2 0.10 solo" "8 0.10 all"; do
  set -- $cfg
  timeout 900 python scripts/edge_cut_p8_probe.py $1 $2 6 $3 > $O/edge_cut_p$1_hot$2_$3.txt 2>&1
  echo "p$1 hot$2 $3 rc=$?" | tee -a $O/status.txt
  tail -2 $O/edge_cut_p$1_hot$2_$3.txt | head -1
done
RAW=/tmp/prof_p8; rm -rf $RAW; mkdir -p $RAW
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $RAW -o p8 -- python $R/scripts/edge_cut_p8_probe.py 8 0.10 6 solo > $O/edge_cut_p8_trace.txt 2>&1)
for f in $(find $RAW -name '*kernel_stats.csv'); do (head -1 $f; grep "glx_\|rocclr" $f) > $O/edge_cut_p8_solo_kernel_stats.csv; done
head -24 $O/edge_cut_p8_solo_kernel_stats.csv
timeout 600 python scripts/filter_bench.py --index > $O/filter_bench_index.txt 2>&1
grep "EdgeWeight\|InDegree\|Topk" $O/filter_bench_index.txt
GLX_DIST_NO_SHORTCUT=1 timeout 600 python bench.py --gpus 1 --force-sharded --steps 20 --warmup 5 --cpu-baseline off --verify > $O/bench_w1_sharded.json 2> $O/bench_w1_sharded.log
python -c "import json; r=json.load(open('$O/bench_w1_sharded.json')); print(r['placements'], r['verified_sharded_equals_unpartitioned'])"
timeout 1500 python -m pytest tests -q -m gpu --timeout 900 > $O/pytest_all.log 2>&1
echo "pytest_all rc=$?" | tee -a $O/status.txt
tail -4 $O/pytest_all.log
