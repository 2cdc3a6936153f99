#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03_run22; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for nb in 512 1024 2048 4096; do
GLX_RESOLVE_BLOCKS=$nb GRAPH_REPLICA=1 MERGED=1 timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof$nb -o p8 --output-format csv -- python $R/scripts/edge_cut_p8_probe.py 8 0.25 6 solo 2>&1 | grep "ONLY rank 0" | cut -c1-120
python - <<PY
import csv,glob
f=glob.glob('$O/prof$nb/*kernel_stats.csv')[0]
for r in csv.DictReader(open(f)):
    if 'resolve' in r['Name'] or 'assign' in r['Name'] or 'finalize' in r['Name']: print($nb, r['Name'][:70], r['Calls'], 'avg us', float(r['AverageNs'])/1e3)
PY
rm -rf $O/prof$nb
done
