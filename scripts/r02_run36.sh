#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=$PWD/gpurun_out/r02_run36
mkdir -p $O
GRAPH_REPLICA=1 timeout 900 python scripts/edge_cut_p8_probe.py 8 0.25 4 > $O/edge_cut_p8_built.txt 2>&1
echo rc=$?
grep -v "^$" $O/edge_cut_p8_built.txt | tail -5 | cut -c1-260
