#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03_run17; mkdir -p $O
cd $R
timeout 600 python bench.py --seed-dist degree --cpu-baseline off --host-boundary off --edge-cut-probe off --small-batches off --other-configs "" > $O/bench_degree.json 2> $O/bench_degree.err; tail -3 $O/bench_degree.err
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r03_run17/bench_degree.json') if l.startswith('{')][-1])
print(d['ms_per_step'], d['value'], d['verified_vs_oracle'], d['config']['seeds'])
r=d['roofline']; print({k:r[k] for k in ('avg_launch_ms','achieved','frac','frac_compulsory','distinct_rows_last_launch')})
print(d['phases'])
PY
timeout 600 python -m pytest tests/test_gpu_dist_ledger.py -m gpu -q -x 2>&1 | tail -3
