R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06/w1; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
GLX_DIST_NO_SHORTCUT=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/w1raw -o t -- python $R/bench.py --gpus 1 --force-sharded --cpu-baseline off --host-boundary off --roofline-probes off --edge-cut-probe off --small-batches off --other-configs "" --pure-leg off --speculate off --design-r off --steps 20 --warmup 3 > $O/bench.json 2> $O/bench.err
f=$(find /tmp/w1raw -name '*kernel_stats.csv'); cp $f $O/kernel_stats.csv
f=$(find /tmp/w1raw -name '*kernel_trace.csv'); (head -1 $f; grep glx_ $f | tail -n 400) > $O/kernel_trace_tail.csv
tail -2 $O/bench.err; head -c 600 $O/bench.json
