#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03_run23; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_dist_store.py tests/test_gpu_dist_ledger.py -m gpu -q -x 2>&1 | tail -5
cd /tmp && export TMPDIR=/tmp
for mode in two one; do
if [ $mode = one ]; then export GLX_RESOLVE_ONE_PASS=1; fi
GRAPH_REPLICA=1 MERGED=1 timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof$mode -o p8 --output-format csv -- python $R/scripts/edge_cut_p8_probe.py 8 0.25 6 solo 2>&1 | grep "ONLY rank 0" | cut -c1-200
python - <<PY
import csv,glob
f=glob.glob('$O/prof$mode/*kernel_stats.csv')[0]
for r in csv.DictReader(open(f)):
    if 'resolve' in r['Name']: print('$mode', r['Name'][:80], r['Calls'], 'avg us', float(r['AverageNs'])/1e3)
PY
rm -rf $O/prof$mode
done
