#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=$PWD/gpurun_out/r02_run38
mkdir -p $O
export TMPDIR=/tmp
T0=$SECONDS
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_n1.json 2> $O/bench_n1.log
echo "bench wall seconds: $((SECONDS-T0))"
python -c "import json; r=json.load(open('$O/bench_n1.json')); print(r['ms_per_step'], r['value'], r['roofline']['frac'], r['host_boundary']['edges_per_s'], r['cpu_baseline']['value']); print(r["small_batches"]); print(r["edge_cut_world1"]["placements"])"
