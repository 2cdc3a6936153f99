#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
show() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', '%.3g'%d['value'], '%.3f ms'%d['ms_per_step'], 'agg2 %.3f'%d['roofline']['avg_launch_ms'], d['config']['pipelined_two_streams'])"; }
python bench.py --cpu-baseline off --pipeline off 2>/dev/null | show n1-seq
python bench.py --cpu-baseline off --pipeline on 2>/dev/null | show n1-pipe
python bench.py --cpu-baseline off --force-sharded --pipeline off 2>/dev/null | show sharded-seq
python bench.py --cpu-baseline off --force-sharded --pipeline on 2>/dev/null | show sharded-pipe
python bench.py --cpu-baseline off --force-sharded --features sharded --pipeline on --steps 5 2>/dev/null | show halo-pipe
