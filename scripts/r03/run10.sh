#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03_run10; mkdir -p $O
cd $R
SECONDS=0; timeout 900 python bench.py --workload c4 --steps 20 --warmup 5 --cpu-baseline off --host-boundary off --edge-cut-probe off --small-batches off --other-configs "" > $O/bench_c4.json 2> $O/bench_c4.err; echo "rc=$?"
echo "wall seconds: $SECONDS"; grep -E "\[bench\]" $O/bench_c4.err | cut -c1-400 | head
rocm-smi --showmeminfo vram 2>/dev/null | head -5
