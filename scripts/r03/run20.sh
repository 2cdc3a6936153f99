#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03_run20; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
GRAPH_REPLICA=1 LEDGER=1 MERGED=1 timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof -o p8 --output-format csv -- python $R/scripts/edge_cut_p8_probe.py 8 0.25 6 2>&1 | grep -v amdgpu.ids | tail -5
ls $O/prof
python - <<PY
import csv,glob
f=glob.glob('$O/prof/*kernel_stats.csv')[0]
rows=list(csv.DictReader(open(f)))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print('total kernel ms', tot/1e6)
for r in rows[:40]:
    print('%-110s %6s %10.3f ms avg %9.1f us' % (r['Name'][:110], r['Calls'], float(r['TotalDurationNs'])/1e6, float(r['AverageNs'])/1e3))
PY
cp $O/prof/*kernel_stats.csv $O/p8_kernel_stats.csv; rm -rf $O/prof
