#!/bin/bash
# Full GPU check: tests + forced-sharded bench paths; logs under gpurun_out/check/.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/check; mkdir -p $O
cd $R
python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; tail -15 $O/pytest.log
python bench.py --workload tiny --batch 8192 --steps 5 --warmup 2 --cpu-baseline off --force-sharded > $O/sharded_repl.json 2> $O/sharded_repl.err; echo "rc=$?"; cut -c1-250 $O/sharded_repl.json; grep -i "error\|Traceback" -A5 $O/sharded_repl.err | head -20
python bench.py --workload tiny --batch 8192 --steps 5 --warmup 2 --cpu-baseline off --force-sharded --features sharded > $O/sharded_halo.json 2> $O/sharded_halo.err; echo "rc=$?"; cut -c1-250 $O/sharded_halo.json; grep -i "error\|Traceback" -A5 $O/sharded_halo.err | head -20
