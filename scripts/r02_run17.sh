#!/bin/bash
# Round 2, GPU run 17: alias pairing loop on wave-uniform state (readlane windows, scalar control).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD
export TMPDIR=/tmp
O=$R/gpurun_out/r02_run17
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_filter.py tests/test_gpu_fuzz.py tests/test_gpu_configs.py -q -m gpu --timeout 600 > $O/pytest_alias.log 2>&1
echo "pytest_alias rc=$?" | tee -a $O/status.txt
tail -4 $O/pytest_alias.log
timeout 600 python scripts/filter_bench.py --index > $O/filter_bench_index.txt 2>&1
grep '"op"' $O/filter_bench_index.txt
RAW=/tmp/prof_h; rm -rf $RAW; mkdir -p $RAW
(cd /tmp && PROBE_SAMPLER=EdgeWeightSampler timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $RAW -o f -- python $R/scripts/filter_hits_probe.py > $O/filter_hits_ew.txt 2>&1)
for f in $(find $RAW -name '*kernel_stats.csv'); do grep "glx_filter\|glx_sample\|glx_alias" $f | cut -c1-220 | tee -a $O/filter_hits_ew_kernels.csv; done
