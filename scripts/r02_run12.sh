#!/bin/bash
# Round 2, GPU run 12: three-stage pipelined edge-cut leg.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD
export TMPDIR=/tmp
O=$R/gpurun_out/r02_run12
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_two_ranks.py tests/test_gpu_dist_store.py tests/test_abi.py -x -q -m gpu > $O/pytest_two.log 2>&1
echo "pytest rc=$?" | tee -a $O/status.txt
tail -4 $O/pytest_two.log
for i in 1 2; do
GLX_DIST_NO_SHORTCUT=1 timeout 600 python bench.py --gpus 1 --force-sharded --steps 20 --warmup 5 --cpu-baseline off --verify > $O/bench_w1_sharded_$i.json 2> $O/bench_w1_sharded_$i.log
python -c "import json; r=json.load(open('$O/bench_w1_sharded_$i.json')); print(r['placements'], r['verified_sharded_equals_unpartitioned'], r['halo_exchange_hop2']['from_replica'])"
done
GLX_DIST_NO_SHORTCUT=1 timeout 600 python bench.py --gpus 1 --force-sharded --steps 20 --warmup 5 --cpu-baseline off --hot-fraction 0.10 > $O/bench_w1_sharded_hot10.json 2> $O/bench_w1_sharded_hot10.log
python -c "import json; r=json.load(open('$O/bench_w1_sharded_hot10.json')); print('hot 0.10', r['placements'])"
HOT_BY=indegree timeout 900 python scripts/edge_cut_p8_probe.py 8 0.25 6 all > $O/edge_cut_p8_hot25.txt 2>&1; tail -3 $O/edge_cut_p8_hot25.txt | cut -c1-300
