#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD
export TMPDIR=/tmp
O=$R/gpurun_out/r02_run25
mkdir -p $O
RAW=/tmp/prof_rs; rm -rf $RAW; mkdir -p $RAW
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $RAW -o rs -- python $R/scripts/resolve_probe.py > $O/resolve_probe.txt 2>&1)
grep "aggregate_begin" $O/resolve_probe.txt | cut -c1-200
for f in $(find $RAW -name '*kernel_stats.csv'); do grep "glx_dist" $f | cut -c1-200 > $O/kernel_stats.csv; done
cat $O/kernel_stats.csv | cut -c1-170
