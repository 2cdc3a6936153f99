#!/bin/bash
# round 3, GPU call 1: new tests, default bench, MFMA ablation (+ its counters)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03_run1; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_round3.py tests/test_gpu_fullsize_oracle.py -m gpu -q -x > $O/pytest_new.log 2>&1; tail -15 $O/pytest_new.log
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"; tail -12 $O/bench_default.err
timeout 300 python scripts/r03/mfma_ablation.py > $O/mfma_ablation_times.txt 2>&1; cat $O/mfma_ablation_times.txt
cd /tmp && export TMPDIR=/tmp
RAW=/tmp/abl; rm -rf $RAW
for c in SQ_INSTS_VALU_MFMA_MOPS_F32 FETCH_SIZE; do
  ABLATION_REPS=2 timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $RAW/$c -o p -- python $R/scripts/r03/mfma_ablation.py > /dev/null 2> $O/abl_$c.err
  f=$(find $RAW/$c -name '*counter_collection.csv' | head -1)
  (head -1 $f; grep glx_aggregate $f) > $O/abl_pmc_$c.csv
done
ls -la $O
