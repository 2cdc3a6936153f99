#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03_run9; mkdir -p $O
cd $R
SECONDS=0; timeout 900 python bench.py --workload c4 --steps 20 --warmup 5 --cpu-baseline off --host-boundary off --edge-cut-probe off --small-batches off --other-configs "" --verify-oracle off > $O/bench_c4.json 2> $O/bench_c4.err; echo "rc=$?"
echo "wall seconds: $SECONDS"; grep -E "\[bench\]" $O/bench_c4.err | head
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03_run9/bench_c4.json').read().strip().splitlines()[-1])
print(d['ms_per_step'], d['value'], d['roofline']['frac'], d['roofline'].get('cache_free'), d.get('roofline_sampler',{}).get('draws_over_uniform_gather_rate'))
PY
