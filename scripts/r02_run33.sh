#!/bin/bash
# Round 2, GPU run 33: final validation of the round; full suite, bench, rocprof + PMC passes for profiles/r02.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD
export TMPDIR=/tmp
O=$R/gpurun_out/r02_run33
mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu --timeout 900 > $O/pytest_all.log 2>&1
echo "pytest_all rc=$?" | tee -a $O/status.txt
tail -4 $O/pytest_all.log
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_n1.json 2> $O/bench_n1.log
python -c "import json; r=json.load(open('$O/bench_n1.json')); print(r['ms_per_step'], r['value'], r['roofline']['avg_launch_ms'], r['roofline']['frac'], r['roofline']['cache_free'], r['host_boundary']['edges_per_s'], r['cpu_baseline']['value'])"
bash scripts/profile_bench.sh r02 > $O/profile.log 2>&1
echo "profile rc=$?" | tee -a $O/status.txt
for B in 1024 8192; do
  timeout 300 python bench.py --batch $B --steps 400 --warmup 40 --cpu-baseline off --roofline-probes off --host-boundary off > $O/bench_b$B.json 2> $O/bench_b$B.log
  python -c "import json; r=json.load(open('$O/bench_b$B.json')); print('B0=$B', r['ms_per_step'], r['value'], r['config']['hipgraph_step'])"
done
for W in c2 c4; do
  timeout 900 python bench.py --workload $W --steps 20 --warmup 5 --cpu-baseline off --roofline-probes off --host-boundary off > $O/bench_$W.json 2> $O/bench_$W.log
  python -c "import json; r=json.load(open('$O/bench_$W.json')); print('$W', r['ms_per_step'], r['value'], r['roofline']['avg_launch_ms'], r['roofline']['frac_algorithmic'])"
done
timeout 900 python bench.py --workload c5 --steps 20 --warmup 5 > $O/bench_c5.json 2> $O/bench_c5.log
python -c "import json; r=json.load(open('$O/bench_c5.json')); print('c5', r['ms_per_step'], r['value'])"
GLX_DIST_NO_SHORTCUT=1 timeout 600 python bench.py --gpus 1 --force-sharded --steps 20 --warmup 5 --cpu-baseline off --verify > $O/bench_w1_sharded.json 2> $O/bench_w1_sharded.log
python -c "import json; r=json.load(open('$O/bench_w1_sharded.json')); print(r['placements'], r['verified_sharded_equals_unpartitioned'])"
