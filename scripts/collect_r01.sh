#!/bin/bash
# Re-collect every round-1 evidence file under gpurun_out/evidence/ (run through gpurun).
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/evidence; mkdir -p $O; cd $R
python bench.py 2>$O/bench_default.err > $O/bench_c3_with_cpu_baseline.json
python bench.py --cpu-baseline off --no-scramble 2>/dev/null > $O/bench_c3_raw_rmat_ids.json
python bench.py --cpu-baseline off --workload c2 2>/dev/null > $O/bench_c2.json
python bench.py --cpu-baseline off --workload c4 --steps 10 --warmup 2 2>/dev/null > $O/bench_c4_single_gpu.json
python bench.py --workload c5 --steps 10 --warmup 2 2>/dev/null > $O/bench_c5_single_gpu.json
python scripts/microbench.py 2>/dev/null > $O/microbench.jsonl
python scripts/reuse_probe.py 2>/dev/null > $O/reuse_probe_raw.txt
( for b in 1024 8192 65536; do python bench.py --cpu-baseline off --batch $b --steps 100 --warmup 20 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(json.dumps({'B0': $b, 'edges_per_s': d['value'], 'ms_per_step': d['ms_per_step'], 'hop2_agg_ms': d['roofline']['avg_launch_ms'], 'frac_of_8TBps': d['roofline']['frac']}))"; done; python scripts/copy_peak.py 2>/dev/null ) > $O/batch_sweep.txt
( for cfg in '1 1024 10' '8 1024 10' '32 1024 10' '32 128 40'; do ./graph-learn_amd/lib/host_path_bench $cfg 21 20000000 256 2>/dev/null | tail -1; done ) > $O/host_path_bench.txt
python scripts/pyapi_bench.py 2>/dev/null | grep -E 'path|init|Loader' > $O/pyapi_bench.txt
python scripts/filter_bench.py 2>/dev/null | grep '^{' > $O/filter_bench.txt
bash scripts/profile_r01.sh r01 > $O/profile.log 2>&1
ls -la $O
