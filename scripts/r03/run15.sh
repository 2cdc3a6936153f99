#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03_run15; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_dist_ledger.py -m gpu -q -x > $O/pytest_ledger.log 2>&1; tail -30 $O/pytest_ledger.log
timeout 900 python -m pytest tests/test_gpu_dist_store.py tests/test_gpu_two_ranks.py tests/test_abi.py -m gpu -q -x > $O/pytest.log 2>&1; tail -30 $O/pytest.log
GLX_DIST_NO_SHORTCUT=1 timeout 600 python bench.py --gpus 1 --force-sharded --cpu-baseline off --host-boundary off --roofline-probes off --edge-cut-probe off --small-batches off --other-configs "" --verify > $O/bench_world1.json 2> $O/bench_world1.err; tail -5 $O/bench_world1.err
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r03_run15/bench_world1.json') if l.startswith('{')][-1])
for k,v in d['placements'].items(): print(k, {a:b for a,b in v.items() if a in ('ms_per_step','count_exchanges_per_step','host_blocked_in_count_exchanges_ms_per_step','ledger')})
print(d['verified_legs'], d['config']['workload'][-200:])
PY
