#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=$PWD/gpurun_out/r02_run32
mkdir -p $O
GLX_RCCL_LIBRARY=$PWD/tests/fake_rccl/libfakerccl.so timeout 900 python tests/scripts/fake_rccl_check.py 2 3 8 > $O/fake_rccl.txt 2>&1
echo rc=$?
tail -30 $O/fake_rccl.txt | cut -c1-220
