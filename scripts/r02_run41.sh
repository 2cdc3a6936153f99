#!/bin/bash
# Round 2, GPU run 41: kernel stats of the world-1 edge-cut run at HEAD.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD
export TMPDIR=/tmp
O=$R/gpurun_out/r02_run41
mkdir -p $O
RAW=/tmp/prof_w1; rm -rf $RAW; mkdir -p $RAW
(cd /tmp && GLX_DIST_NO_SHORTCUT=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $RAW -o w1 -- python $R/bench.py --gpus 1 --force-sharded --steps 20 --warmup 5 --cpu-baseline off --features sharded > $O/bench.json 2> $O/bench.log)
for f in $(find $RAW -name '*kernel_stats.csv'); do (head -1 $f; grep "glx_\|rccl\|nccl" $f) | cut -c1-260 > $O/kernel_stats_world1_edge_cut.csv; done
head -16 $O/kernel_stats_world1_edge_cut.csv | cut -c1-170
