#!/bin/bash
# rocprofv3 passes for the bench (run on the GPU box through gpurun).
# Raw outputs stay in /tmp; only small summaries go to gpurun_out/prof_<tag>/.
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_$TAG
RAW=/tmp/prof_raw
rm -rf $RAW; mkdir -p $OUT $RAW
cd /tmp && export TMPDIR=/tmp
BENCH="python $R/bench.py --cpu-baseline off"
rocprofv3 --kernel-trace --stats --output-format csv -d $RAW/trace -o trace -- $BENCH --steps 20 --warmup 5 --roofline-probes off > $OUT/bench_trace.json 2> $OUT/bench_trace.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $RAW/fetch -o fetch -- $BENCH --steps 5 --warmup 1 > $OUT/bench_fetch.json 2> $OUT/bench_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $RAW/write -o write -- $BENCH --steps 5 --warmup 1 > $OUT/bench_write.json 2> $OUT/bench_write.err
find $RAW -type f | head -40
for f in $(find $RAW/trace -name '*stats*.csv'); do cp $f $OUT/; done
# per-dispatch rows of the glx kernels only (torch's generator kernels are noise)
for d in trace fetch write; do
  for f in $(find $RAW/$d -name '*kernel_trace.csv' -o -name '*counter_collection.csv'); do
    b=$(basename $f)
    (head -1 $f; grep glx_ $f) > $OUT/${d}_$b
  done
done
tail -3 $OUT/*.err
ls -la $OUT
