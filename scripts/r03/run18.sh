#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03_run18; mkdir -p $O
cd $R
for p in off on; do
for rep in 1 2; do
timeout 600 python bench.py --pipeline $p --steps 40 --cpu-baseline off --host-boundary off --edge-cut-probe off --small-batches off --other-configs "" --roofline-probes off --verify-oracle off 2> /dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('pipeline $p', d['ms_per_step'], d['value'])"
done; done
