#!/bin/bash
# Round 2, GPU run 37: full suite + smoke + the driver's bench command at HEAD.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD
export TMPDIR=/tmp
O=$R/gpurun_out/r02_run37
mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu --timeout 900 > $O/pytest_all.log 2>&1
echo "pytest_all rc=$?" | tee -a $O/status.txt
grep -n "passed\|failed" $O/pytest_all.log | tail -2
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1
echo "smoke rc=$?" | tee -a $O/status.txt
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_n1.json 2> $O/bench_n1.log
python -c "import json; r=json.load(open('$O/bench_n1.json')); print(r['ms_per_step'], r['value'], r['roofline']['frac'], r['host_boundary']['edges_per_s'], r['cpu_baseline']['value'])"
