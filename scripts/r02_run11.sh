#!/bin/bash
# Round 2, GPU run 11: hot rows ranked by access count: coverage and per-rank work at P = 8.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD
export TMPDIR=/tmp
O=$R/gpurun_out/r02_run11
mkdir -p $O
for cfg in "8 0.10 all" "8 0.25 all" "8 0.05 all"; do
  set -- $cfg
  HOT_BY=access timeout 900 python scripts/edge_cut_p8_probe.py $1 $2 6 $3 > $O/edge_cut_access_p$1_hot$2_$3.txt 2>&1
  echo "p$1 hot$2 $3 rc=$?" | tee -a $O/status.txt
  tail -3 $O/edge_cut_access_p$1_hot$2_$3.txt | cut -c1-330
done
GLX_DIST_NO_SHORTCUT=1 timeout 600 python bench.py --gpus 1 --force-sharded --steps 20 --warmup 5 --cpu-baseline off --verify > $O/bench_w1_sharded.json 2> $O/bench_w1_sharded.log
python -c "import json; r=json.load(open('$O/bench_w1_sharded.json')); print(r['placements'], r['verified_sharded_equals_unpartitioned'], r['halo_exchange_hop2'])"
grep "hot-row" $O/bench_w1_sharded.log
timeout 600 python -m pytest tests/test_gpu_two_ranks.py -x -q -m gpu > $O/pytest_two.log 2>&1; tail -3 $O/pytest_two.log
