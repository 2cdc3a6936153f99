#!/bin/bash
# Round 2, GPU run 20: graph replica (hot vertices' adjacency rows on every GPU): tests, P = 8 on one GPU with / without.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD
export TMPDIR=/tmp
O=$R/gpurun_out/r02_run20
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_dist_store.py tests/test_gpu_two_ranks.py -q -m gpu --timeout 600 > $O/pytest_dist.log 2>&1
echo "pytest_dist rc=$?" | tee -a $O/status.txt
tail -5 $O/pytest_dist.log
for G in 0 1; do
  GRAPH_REPLICA=$G timeout 900 python scripts/edge_cut_p8_probe.py 8 0.25 6 > $O/edge_cut_p8_hot25_replica$G.txt 2>&1
  echo "p8 replica=$G rc=$?" | tee -a $O/status.txt
  grep -v "^$" $O/edge_cut_p8_hot25_replica$G.txt | tail -5
done
GLX_DIST_NO_SHORTCUT=1 timeout 600 python bench.py --gpus 1 --force-sharded --steps 20 --warmup 5 --cpu-baseline off --verify > $O/bench_w1_sharded.json 2> $O/bench_w1_sharded.log
python -c "import json; r=json.load(open('$O/bench_w1_sharded.json')); print(r['placements'], r['verified_sharded_equals_unpartitioned'], r['sampling_exchange_hop2'])"
tail -3 $O/bench_w1_sharded.log
