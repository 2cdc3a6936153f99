#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/prof_sharded; RAW=/tmp/prof_raw_s
rm -rf $RAW; mkdir -p $OUT $RAW; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $RAW -o t -- python $R/bench.py --cpu-baseline off --force-sharded --pipeline off --steps 10 --warmup 2 > $OUT/bench.json 2> $OUT/bench.err
cp $RAW/t_kernel_stats.csv $OUT/ 2>/dev/null || find $RAW -name '*kernel_stats.csv' -exec cp {} $OUT/ \;
python - <<'PY'
import csv,glob,os
f=glob.glob(os.environ.get('OUT','/root/repo/gpurun_out/prof_sharded')+'/*kernel_stats.csv')[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:-float(r['TotalDurationNs']))
for r in rows[:28]: print("%-90s %6s %10.3f ms %8.1f us"%(r['Name'][:90], r['Calls'], float(r['TotalDurationNs'])/1e6, float(r['AverageNs'])/1e3))
PY
