#!/bin/bash
# Round 2, GPU run 10: aggregate begin/end split (halo prefetch on the sampling stream), global in-degree flag.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD
export TMPDIR=/tmp
O=$R/gpurun_out/r02_run10
mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu --timeout 900 > $O/pytest_all.log 2>&1
echo "pytest_all rc=$?" | tee -a $O/status.txt
tail -6 $O/pytest_all.log
for i in 1 2; do
GLX_DIST_NO_SHORTCUT=1 timeout 600 python bench.py --gpus 1 --force-sharded --steps 20 --warmup 5 --cpu-baseline off --verify > $O/bench_w1_sharded_$i.json 2> $O/bench_w1_sharded_$i.log
python -c "import json; r=json.load(open('$O/bench_w1_sharded_$i.json')); print(r['placements'], r['verified_sharded_equals_unpartitioned'])"
done
GLX_DIST_NO_SHORTCUT=1 timeout 600 python bench.py --gpus 1 --force-sharded --steps 20 --warmup 5 --cpu-baseline off --pipeline off > $O/bench_w1_sharded_nopipe.json 2> $O/bench_w1_sharded_nopipe.log
python -c "import json; r=json.load(open('$O/bench_w1_sharded_nopipe.json')); print('no pipeline', r['placements'])"
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_n1.json 2> $O/bench_n1.log
python -c "import json; r=json.load(open('$O/bench_n1.json')); print(r['ms_per_step'], r['roofline']['frac'], r.get('host_boundary'), r['cpu_baseline']['value'])"
