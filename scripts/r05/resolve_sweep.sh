#!/bin/bash
# Kernel time of glx_dist_resolve_kernel on the world-1 partitioned headline workload under the grid / ids-per-thread knobs
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r05_w1; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for cfg in "2 1024" "2 2048" "2 4096" "4 1024" "4 2048" "4 4096" "8 1024" "8 2048"; do
  set -- $cfg
  GLX_RESOLVE_IDS=$1 GLX_RESOLVE_BLOCKS=$2 GLX_DIST_NO_SHORTCUT=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rs_$1_$2 -o t -- python $R/bench.py --gpus 1 --force-sharded --cpu-baseline off --host-boundary off --roofline-probes off --edge-cut-probe off --small-batches off --other-configs "" --pure-leg off --speculate off --design-r off --steps 20 --warmup 3 > /tmp/rs_$1_$2.json 2> /tmp/rs_$1_$2.err
  f=$(find /tmp/rs_$1_$2 -name '*kernel_stats.csv')
  echo "ids=$1 blocks=$2 $(grep resolve_kernel $f | awk -F, '{print "calls="$(NF-6)" avg_ns="$(NF-4)}') $(python -c "
import json;d=json.loads(open('/tmp/rs_$1_$2.json').read().splitlines()[-1]);print({k:v['ms_per_step'] for k,v in d['placements'].items()})")" | tee -a $O/resolve_sweep.txt
done
