#!/bin/bash
# Round 2, GPU run 2: whole GPU suite incl. the new full-size config tests, rocprof passes, probes.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD
export TMPDIR=/tmp
O=$R/gpurun_out/r02_run2
mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu --timeout 900 --durations=15 > $O/pytest_all.log 2>&1
echo "pytest_all rc=$?" | tee -a $O/status.txt
tail -40 $O/pytest_all.log
bash scripts/profile_bench.sh r02 > $O/profile.log 2>&1
echo "profile rc=$?" | tee -a $O/status.txt
# kernel trace of the edge-cut path (world size 1 over RCCL, generic path)
RAW=/tmp/prof_w1; rm -rf $RAW; mkdir -p $RAW
(cd /tmp && GLX_DIST_NO_SHORTCUT=1 rocprofv3 --kernel-trace --stats --output-format csv -d $RAW -o w1 -- python $R/bench.py --gpus 1 --force-sharded --steps 20 --warmup 5 --cpu-baseline off > $O/bench_w1_trace.json 2> $O/bench_w1_trace.err)
for f in $(find $RAW -name '*stats*.csv'); do cp $f $O/w1_$(basename $f); done
echo "w1 trace rc=$?" | tee -a $O/status.txt
timeout 600 python scripts/hot_layout_probe.py > $O/hot_layout_probe.txt 2>&1
echo "hot probe rc=$?" | tee -a $O/status.txt
cat $O/hot_layout_probe.txt | tail -8
for B in 1024 8192; do
  timeout 300 python bench.py --batch $B --steps 200 --warmup 20 --cpu-baseline off --roofline-probes off > $O/bench_b$B.json 2> $O/bench_b$B.log
  python -c "import json; r=json.load(open('$O/bench_b$B.json')); print('B0=$B', r['ms_per_step'], r['value'])"
done
