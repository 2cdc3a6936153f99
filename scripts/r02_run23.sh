#!/bin/bash
# Round 2, GPU run 23: replica rows answer straight into the response; full suite; world-1 edge-cut bench.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD
export TMPDIR=/tmp
O=$R/gpurun_out/r02_run23
mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu --timeout 900 > $O/pytest_all.log 2>&1
echo "pytest_all rc=$?" | tee -a $O/status.txt
grep -n "passed\|failed" $O/pytest_all.log | tail -2
for i in 1 2; do
GLX_DIST_NO_SHORTCUT=1 timeout 600 python bench.py --gpus 1 --force-sharded --steps 20 --warmup 5 --cpu-baseline off --verify > $O/bench_w1_sharded_$i.json 2> $O/bench_w1_sharded_$i.log
python -c "import json; r=json.load(open('$O/bench_w1_sharded_$i.json')); print(r['placements'], r['verified_sharded_equals_unpartitioned'], r['sampling_exchange_hop2']['from_graph_replica'])"
done
GLX_DIST_NO_SHORTCUT=1 timeout 600 python bench.py --gpus 1 --force-sharded --steps 20 --warmup 5 --cpu-baseline off --graph-replica off > $O/bench_w1_sharded_noreplica.json 2> $O/bench_w1_sharded_noreplica.log
python -c "import json; r=json.load(open('$O/bench_w1_sharded_noreplica.json')); print('no graph replica', r['placements'])"
