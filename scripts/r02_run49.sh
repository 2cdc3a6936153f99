#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=$PWD/gpurun_out/r02_run49
mkdir -p $O
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -DGLX_ALIAS_PROFILE -I include -I graph-learn_amd/csrc scripts/probes/alias_row_probe.hip -o /tmp/alias_row_probe 2> $O/compile.log
for n in 138719 20000; do /tmp/alias_row_probe $n | tail -1; done | tee $O/alias_row_probe.txt
