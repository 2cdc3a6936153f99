#!/bin/bash
# Round 2, GPU run 15: column-slice A/B of the aggregate kernel; hit distribution + kernel breakdown of id == value filters.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD
export TMPDIR=/tmp
O=$R/gpurun_out/r02_run15
mkdir -p $O
timeout 600 python scripts/agg_slices_probe.py > $O/agg_slices_probe.txt 2>&1
cat $O/agg_slices_probe.txt | grep round
for S in TopkSampler RandomWithoutReplacementSampler; do
  RAW=/tmp/prof_h_$S; rm -rf $RAW; mkdir -p $RAW
  (cd /tmp && PROBE_SAMPLER=$S timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $RAW -o f -- python $R/scripts/filter_hits_probe.py > $O/filter_hits_$S.txt 2>&1)
  grep "^rows" $O/filter_hits_$S.txt
  for f in $(find $RAW -name '*kernel_stats.csv'); do grep "glx_filter\|glx_sample\|glx_alias" $f | cut -c1-220 | tee -a $O/filter_hits_${S}_kernels.csv; done
done
