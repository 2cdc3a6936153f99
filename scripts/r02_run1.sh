#!/bin/bash
# Round 2, GPU run 1: new distributed-store tests first, then the whole GPU suite, then bench legs.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r02_run1
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_dist_store.py tests/test_host_cpp.py tests/test_gpu_sharded.py -x -q -m gpu > $O/pytest_new.log 2>&1
echo "pytest_new rc=$?" | tee -a $O/status.txt
tail -5 $O/pytest_new.log
timeout 1200 python -m pytest tests -x -q -m gpu > $O/pytest_all.log 2>&1
echo "pytest_all rc=$?" | tee -a $O/status.txt
tail -5 $O/pytest_all.log
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_n1.json 2> $O/bench_n1.log
echo "bench_n1 rc=$?" | tee -a $O/status.txt
cat $O/bench_n1.json | head -c 3000
for U in 10 12; do
  GLX_AGG_UNROLL=$U timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --cpu-baseline off --roofline-probes off > $O/bench_n1_unroll$U.json 2> $O/bench_n1_unroll$U.log
  echo "bench unroll $U rc=$?" | tee -a $O/status.txt
done
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --cpu-baseline off --roofline-probes off --pipeline on > $O/bench_n1_pipeline.json 2> $O/bench_n1_pipeline.log
echo "bench pipeline rc=$?" | tee -a $O/status.txt
# the edge-cut path over RCCL with one rank: generic path (no world-size-1 shortcut), hot-row replica 10 %
GLX_DIST_NO_SHORTCUT=1 timeout 600 python bench.py --gpus 1 --force-sharded --steps 20 --warmup 5 --cpu-baseline off --verify > $O/bench_w1_sharded.json 2> $O/bench_w1_sharded.log
echo "bench w1 sharded rc=$?" | tee -a $O/status.txt
tail -3 $O/bench_w1_sharded.log
cat $O/bench_w1_sharded.json | head -c 3000
grep -h '"ms_per_step"' $O/*.json | python -c "
import sys, json
for ln in sys.stdin:
    try:
        r = json.loads(ln)
        print(r['config']['workload'][:40], r['ms_per_step'], r['roofline']['avg_launch_ms'], r.get('placements'))
    except Exception as e:
        print('bad line', e)
"
