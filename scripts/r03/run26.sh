#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03_run26; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for m in two one; do
if [ $m = one ]; then export GLX_RESOLVE_ONE_PASS=1; else unset GLX_RESOLVE_ONE_PASS; fi
GRAPH_REPLICA=1 LEDGER=1 MERGED=1 timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof$m -o p8 --output-format csv -- python $R/scripts/edge_cut_p8_probe.py 8 0.25 6 2>&1 | grep -E "^P = " | cut -c48-130
python - <<PY
import csv,glob
f=glob.glob('$O/prof$m/*kernel_stats.csv')[0]
tot=0
for r in csv.DictReader(open(f)):
    n=r['Name']
    if any(k in n for k in ('glx_dist_','glx_lookup','aggregate_kernel','sample_slots','copyBuffer','glx_part_')):
        print('$m %-72s calls %5s total %9.3f ms avg %9.1f us' % (n[:72], r['Calls'], float(r['TotalDurationNs'])/1e6, float(r['AverageNs'])/1e3))
PY
rm -rf $O/prof$m
done
