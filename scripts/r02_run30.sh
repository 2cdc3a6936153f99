#!/bin/bash
# Round 2, GPU run 30: host boundary rate by thread count and requests per thread.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=$PWD/gpurun_out/r02_run30
mkdir -p $O
for rep in 1 2; do
for T in 8 12 16 24 32; do
  ./graph-learn_amd/lib/host_path_bench $T 1024 10 2>&1 | grep "^{" | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('threads', r['threads'], 'reps 10', r['sampled_edges_per_s_host_pointer_path'])"
done
done | tee $O/host_path_threads.txt
for T in 16 32; do
  ./graph-learn_amd/lib/host_path_bench $T 1024 40 2>&1 | grep "^{" | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('threads', r['threads'], 'reps 40', r['sampled_edges_per_s_host_pointer_path'])"
done | tee -a $O/host_path_threads.txt
