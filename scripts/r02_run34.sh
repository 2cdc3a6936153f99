#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=$PWD/gpurun_out/r02_run34
mkdir -p $O
timeout 600 python scripts/graph_replica_tradeoff.py 2>/dev/null | tee $O/graph_replica_tradeoff.txt
