#!/bin/bash
# AddressSanitizer build of the C++ host mirror + its unit test programs (the device library stays as it is).
#   bash scripts/asan_host.sh build        here (no GPU needed): binaries under graph-learn_amd/lib/asan/
#   bash scripts/asan_host.sh run          on a GPU box: runs them, exit code 0 = no report
set -e
R=$(cd "$(dirname "$0")/.." && pwd)/graph-learn_amd
SAN=${SAN:-address}   # SAN=undefined: UndefinedBehaviorSanitizer instead
O=$R/lib/san_$SAN
if [ "$1" = build ]; then
  mkdir -p $O
  for t in sampler_unittest aggregating_op_unittest partition_stitch_unittest graph_op_unittest request_unittest loader_unittest dag_unittest; do
    g++ -std=c++17 -O1 -g -fsanitize=$SAN -fno-sanitize-recover=all -fno-omit-frame-pointer -fPIC -pthread -I$R/../include -I$R/host/include -I$R/host/test \
      $R/host/src/*.cc $R/host/test/$t.cpp -o $O/$t -L$R/lib -lglx -Wl,-rpath,'$ORIGIN/..' -Wl,-rpath,/opt/rocm/lib &
  done
  wait
  ls $O
else
  rc=0
  for t in $(ls $O); do
    d=$(mktemp -d); c=0; ( cd $d && ASAN_OPTIONS="detect_leaks=0:protect_shadow_gap=0:abort_on_error=0:exitcode=77" timeout 600 $O/$t > $d/out.txt 2>&1 ) || c=$?
    echo "$t rc=$c $(tail -1 $d/out.txt | cut -c1-120)"
    if [ $c -ne 0 ]; then rc=1; grep -m1 -A25 "ERROR: AddressSanitizer" $d/out.txt | cut -c1-200; fi
  done
  exit $rc
fi
