#!/bin/bash
# Round 2, GPU run 31: full suite after the partition / alias-test changes; N = 1 bench; world-1 edge-cut bench.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD
export TMPDIR=/tmp
O=$R/gpurun_out/r02_run31
mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu --timeout 900 > $O/pytest_all.log 2>&1
echo "pytest_all rc=$?" | tee -a $O/status.txt
grep -n "passed\|failed\|Error" $O/pytest_all.log | tail -5
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1
echo "smoke rc=$?" | tee -a $O/status.txt
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_n1.json 2> $O/bench_n1.log
python -c "import json; r=json.load(open('$O/bench_n1.json')); print(r['ms_per_step'], r['value'], r['roofline']['frac'], r['host_boundary']['edges_per_s'], r['cpu_baseline']['value'])"
GLX_DIST_NO_SHORTCUT=1 timeout 600 python bench.py --gpus 1 --force-sharded --steps 20 --warmup 5 --cpu-baseline off --verify > $O/bench_w1.json 2> $O/bench_w1.log
python -c "import json; r=json.load(open('$O/bench_w1.json')); print(r['placements'], r['verified_sharded_equals_unpartitioned'])"
