#!/bin/bash
# Round 2, GPU run 35: the graph replica built from the shards (collective), C++ runner, Python SPMD, fake-rccl, bench.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD
export TMPDIR=/tmp
O=$R/gpurun_out/r02_run35
mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_dist_store.py tests/test_gpu_two_ranks.py tests/test_host_cpp.py tests/test_gpu_pyapi.py -q -m gpu --timeout 900 > $O/pytest.log 2>&1
echo "pytest rc=$?" | tee -a $O/status.txt
grep -n "passed\|failed" $O/pytest.log | tail -2
GLX_DIST_NO_SHORTCUT=1 timeout 600 python bench.py --gpus 1 --force-sharded --steps 20 --warmup 5 --cpu-baseline off --verify > $O/bench_w1.json 2> $O/bench_w1.log
python -c "import json; r=json.load(open('$O/bench_w1.json')); print(r['placements'], r['verified_sharded_equals_unpartitioned'], r['sampling_exchange_hop2'])"
grep "graph replica" $O/bench_w1.log
