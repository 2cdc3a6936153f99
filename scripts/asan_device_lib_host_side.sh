#!/bin/bash
# AddressSanitizer over the HOST code of libglx.so (scratch arenas, request carving, the distributed store's bookkeeping,
# handle lifetimes): csrc/*.hip compiled with -fsanitize=address -fno-gpu-sanitize (device code untouched), the host
# mirror and its unit test programs with the same compiler and runtime.
#   bash scripts/asan_device_lib_host_side.sh build     here (cross-compiles)
#   bash scripts/asan_device_lib_host_side.sh run       on a GPU box; exit code 0 = no report
# (The Python GPU tests cannot run this way: with the sanitizer runtime preloaded into an interpreter that also holds
# torch's bundled HIP runtime, the runtime's hsa_amd_memory_pool_allocate interceptor fails every device allocation.)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
R=$ROOT/graph-learn_amd
O=$R/lib/asan_hip
CXX=/opt/rocm/lib/llvm/bin/clang++
if [ "$1" = build ]; then
  mkdir -p $O/obj
  for f in $R/csrc/*.hip; do
    b=$(basename $f .hip)
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O1 -g -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt \
      -fsanitize=address -fno-gpu-sanitize -shared-libsan -fno-omit-frame-pointer -Wno-unused-result -I$ROOT/include -I$R/csrc -c $f -o $O/obj/$b.o &
  done
  wait
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -fsanitize=address -fno-gpu-sanitize -shared-libsan -o $O/libglx.so $O/obj/*.o
  for t in sampler_unittest aggregating_op_unittest partition_stitch_unittest graph_op_unittest; do
    $CXX -std=c++17 -O1 -g -fsanitize=address -shared-libsan -fno-omit-frame-pointer -fPIC -pthread -I$ROOT/include -I$R/host/include -I$R/host/test \
      $R/host/src/*.cc $R/host/test/$t.cpp -o $O/$t -L$O -lglx -Wl,-rpath,'$ORIGIN' -Wl,-rpath,/opt/rocm/lib \
      -Wl,-rpath,$(dirname $($CXX -print-file-name=libclang_rt.asan-x86_64.so)) &
  done
  wait
  rm -rf $O/obj
  ls -la $O
else
  rc=0
  for t in sampler_unittest aggregating_op_unittest partition_stitch_unittest graph_op_unittest; do
    d=$(mktemp -d); c=0
    ( cd $d && ASAN_OPTIONS="detect_leaks=0:protect_shadow_gap=0:abort_on_error=0:exitcode=77" timeout 600 $O/$t > $d/out.txt 2>&1 ) || c=$?
    echo "$t rc=$c $(tail -1 $d/out.txt | cut -c1-120)"
    mkdir -p $ROOT/gpurun_out/asan_hip; cp $d/out.txt $ROOT/gpurun_out/asan_hip/$t.txt
    # exit code 77 with "CHECK failed: sanitizer_allocator_device.h ... dev_runtime_unloaded_" and no "ERROR:" line is
    # the sanitizer runtime's own assertion while libhsa-runtime64 frees memory in __cxa_finalize after the device runtime
    # went away (intermittent, after the last test passed): not a finding about this code
    if grep -q "ERROR: AddressSanitizer" $d/out.txt || ! grep -q "test(s), 0 failure(s)" $d/out.txt; then
      rc=1; grep -m1 -A30 "ERROR: AddressSanitizer" $d/out.txt | cut -c1-220; tail -5 $d/out.txt | cut -c1-220
    fi
  done
  exit $rc
fi
