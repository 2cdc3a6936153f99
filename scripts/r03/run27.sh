#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03_run27; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_dist_store.py tests/test_gpu_dist_ledger.py tests/test_gpu_sharded.py -m gpu -q -x > $O/pytest.log 2>&1; grep -E "passed|failed|error" $O/pytest.log | tail -3
for rep in 1 2; do
for m in queue noqueue; do
if [ $m = noqueue ]; then export GLX_RESOLVE_NO_QUEUE=1; else unset GLX_RESOLVE_NO_QUEUE; fi
echo -n "$m: "; GRAPH_REPLICA=1 LEDGER=1 MERGED=1 timeout 600 python scripts/edge_cut_p8_probe.py 8 0.25 10 2>&1 | grep -E "^P = " | cut -c48-130
done; done
cd /tmp && export TMPDIR=/tmp
for m in queue noqueue; do
if [ $m = noqueue ]; then export GLX_RESOLVE_NO_QUEUE=1; else unset GLX_RESOLVE_NO_QUEUE; fi
GRAPH_REPLICA=1 MERGED=1 timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof$m -o p8 --output-format csv -- python $R/scripts/edge_cut_p8_probe.py 8 0.25 6 solo 2>&1 | grep "ONLY rank 0" | cut -c48-140
python - <<PY
import csv,glob
f=glob.glob('$O/prof$m/*kernel_stats.csv')[0]
for r in csv.DictReader(open(f)):
    if 'resolve' in r['Name']: print('$m', r['Name'][:80], r['Calls'], 'avg us', float(r['AverageNs'])/1e3)
PY
rm -rf $O/prof$m
done
