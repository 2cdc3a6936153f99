#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03_run21; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
GRAPH_REPLICA=1 LEDGER=1 MERGED=1 timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof -o p8 --output-format csv -- python $R/scripts/edge_cut_p8_probe.py 8 0.25 6 solo 2>&1 | grep -v "amdgpu.ids\|rocprofv3\|output_stream" | tail -6
python - <<PY
import csv,glob
f=glob.glob('$O/prof/*kernel_trace.csv')[0]
rows=list(csv.DictReader(open(f)))
# the last 6 steps: find the aggregate<...,3> launches with big grids
rows.sort(key=lambda r:int(r['Start_Timestamp']))
aggs=[i for i,r in enumerate(rows) if 'glx_aggregate_kernel' in r['Kernel_Name'] and ', 3>' in r['Kernel_Name'] and int(r['Grid_Size_X'])>1000000]
print('big 3-source aggregates:', len(aggs))
# window: from the first kernel after the 3rd-last hop-1 aggregate ... use last 4 steps: between big aggregates
starts=[int(rows[i]['Start_Timestamp']) for i in aggs]
import collections
lo=starts[-5]; hi=starts[-1]
acc=collections.Counter(); cnt=collections.Counter()
for r in rows:
    s=int(r['Start_Timestamp'])
    if lo<=s<hi:
        n=r['Kernel_Name'][:90]
        acc[n]+=int(r['End_Timestamp'])-s; cnt[n]+=1
print('wall per step over 4 steps: %.3f ms; kernel sum per step %.3f ms' % ((hi-lo)/4e6, sum(acc.values())/4e6))
for n,v in acc.most_common(25): print('%-92s %4d %8.3f ms/step' % (n, cnt[n]/4, v/4e6))
PY
cp $O/prof/*kernel_stats.csv $O/p8_solo_kernel_stats.csv; rm -rf $O/prof
