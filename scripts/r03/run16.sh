#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03_run16; mkdir -p $O
cd $R
for mode in "0 0" "0 1" "1 1"; do
  set -- $mode
  echo "=== LEDGER=$1 MERGED=$2" >> $O/p8.txt
  GRAPH_REPLICA=1 LEDGER=$1 MERGED=$2 timeout 600 python scripts/edge_cut_p8_probe.py 8 0.25 6 2>&1 | grep -v amdgpu.ids >> $O/p8.txt
done
cat $O/p8.txt | cut -c1-400
