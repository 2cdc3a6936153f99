#!/bin/bash
# Round 2, GPU run 29: 8 ranks as threads on one GPU after the graph replica / rank-select / partition work.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD
export TMPDIR=/tmp
O=$R/gpurun_out/r02_run29
mkdir -p $O
GRAPH_REPLICA=1 timeout 900 python scripts/edge_cut_p8_probe.py 8 0.25 6 > $O/edge_cut_p8_all.txt 2>&1
grep -v "^$" $O/edge_cut_p8_all.txt | tail -4 | cut -c1-250
GRAPH_REPLICA=1 timeout 900 python scripts/edge_cut_p8_probe.py 8 0.25 6 solo > $O/edge_cut_p8_solo.txt 2>&1
grep -v "^$" $O/edge_cut_p8_solo.txt | tail -3 | cut -c1-250
GRAPH_REPLICA=1 timeout 900 python scripts/edge_cut_p8_probe.py 2 0.25 6 > $O/edge_cut_p2_all.txt 2>&1
grep -v "^$" $O/edge_cut_p2_all.txt | tail -3 | cut -c1-200
