#!/bin/bash
# Round 2, GPU run 18: where one long row's alias build spends its cycles; parity of the tables.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD
O=$R/gpurun_out/r02_run18
mkdir -p $O
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -DGLX_ALIAS_PROFILE -I include -I graph-learn_amd/csrc scripts/probes/alias_row_probe.hip -o /tmp/alias_row_probe 2> $O/compile.log
for n in 138719 20000 1000; do /tmp/alias_row_probe $n | tail -1; done | tee $O/alias_row_probe.txt
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_filter.py tests/test_gpu_fuzz.py tests/test_gpu_configs.py -q -m gpu --timeout 600 > $O/pytest_alias.log 2>&1
echo "pytest_alias rc=$?" | tee -a $O/status.txt
tail -4 $O/pytest_alias.log
timeout 600 python scripts/filter_bench.py --index > $O/filter_bench_index.txt 2>&1
grep '"op"' $O/filter_bench_index.txt
