cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$PWD; export TMPDIR=/tmp
for W in ts id; do
RAW=/tmp/prof_f_$W; rm -rf $RAW; mkdir -p $RAW
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $RAW -o f -- python $R/scripts/filter_alias_probe.py $W > /dev/null 2>&1)
echo "== $W"; for f in $(find $RAW -name '*kernel_stats.csv'); do grep "glx_filter\|glx_sample\|glx_alias" $f | cut -c1-200; done
done
