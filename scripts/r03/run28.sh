#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03_run28; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for m in flush; do
if [ $m = noflush ]; then export GLX_RESOLVE_NO_FLUSH=1; else unset GLX_RESOLVE_NO_FLUSH; fi
GRAPH_REPLICA=1 MERGED=1 timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof$m -o p8 --output-format csv -- python $R/scripts/edge_cut_p8_probe.py 8 0.25 6 solo 2>&1 | grep "ONLY rank 0" | cut -c48-140
python - <<PY
import csv,glob
f=glob.glob('$O/prof$m/*kernel_stats.csv')[0]
for r in csv.DictReader(open(f)):
    if 'resolve' in r['Name']: print('$m', r['Name'][:80], r['Calls'], 'avg us', float(r['AverageNs'])/1e3)
PY
rm -rf $O/prof$m
done
