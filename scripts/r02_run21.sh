#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=$PWD/gpurun_out/r02_run21
mkdir -p $O /tmp/spmd
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29577 scripts/spmd_pyapi_check.py /tmp/spmd --share-device > $O/spmd.out 2> $O/spmd.err
echo rc=$?
grep -v "^\[Gloo\]\|^$" $O/spmd.err | grep -B 30 "Error\|error" | head -80
