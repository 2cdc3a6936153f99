#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for rep in 1 2 3; do
for m in two one; do
if [ $m = one ]; then export GLX_RESOLVE_ONE_PASS=1; else unset GLX_RESOLVE_ONE_PASS; fi
echo -n "$m: "; GRAPH_REPLICA=1 LEDGER=1 MERGED=1 timeout 600 python scripts/edge_cut_p8_probe.py 8 0.25 10 2>&1 | grep -E "^P = " | cut -c48-130
done; done
