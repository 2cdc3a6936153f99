#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03_run7; mkdir -p $O
cd $R
timeout 1200 python -m pytest tests/test_gpu_dist_store.py tests/test_gpu_two_ranks.py tests/test_gpu_sharded.py tests/test_host_cpp.py -m gpu -q -x > $O/pytest.log 2>&1; tail -30 $O/pytest.log
timeout 900 python bench.py --steps 20 --warmup 5 --cpu-baseline off --other-configs "" --host-boundary off --small-batches off > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -5 $O/bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03_run7/bench.json').read().strip().splitlines()[-1])
print(d['ms_per_step'], json.dumps(d['edge_cut_world1'])[:1800])
PY
