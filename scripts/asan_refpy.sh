#!/bin/bash
# The reference's own Python test files (tests/refpy.py) on an AddressSanitizer build of the extension + host mirror.
#   bash scripts/asan_refpy.sh build     here: graph-learn_amd/lib/asan_py/pywrap_graphlearn*.so (host sources compiled in)
#   bash scripts/asan_refpy.sh run       on a GPU box: every staged test file under LD_PRELOAD=libasan; logs in gpurun_out/refpy/
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
R=$ROOT/graph-learn_amd
O=$R/lib/asan_py
EXT=$(python3 -c "import sysconfig; print(sysconfig.get_config_var('EXT_SUFFIX'))")
if [ "$1" = build ]; then
  mkdir -p $O
  g++ -std=c++17 -O1 -g -fsanitize=address -fno-omit-frame-pointer -fPIC -pthread -fvisibility=hidden -I$R/../include -I$R/host/include \
    $(python3 -m pybind11 --includes) -shared -o $O/pywrap_graphlearn$EXT $R/python/pywrap_graphlearn.cc $R/host/src/*.cc -L$R/lib -lglx
  ls -la $O
else
  export GLX_REFPY_MODULE=$O/pywrap_graphlearn$EXT
  export GLX_REFPY_PRELOAD=$(gcc -print-file-name=libasan.so)
  export ASAN_OPTIONS="detect_leaks=0:protect_shadow_gap=0:abort_on_error=0:exitcode=77:log_path=$ROOT/gpurun_out/refpy/asan"
  mkdir -p $ROOT/gpurun_out/refpy
  python3 $ROOT/tests/scripts/refpy_run_all.py "${@:2}"
  ls $ROOT/gpurun_out/refpy/asan.* 2>/dev/null | head -5 || true
  if ls $ROOT/gpurun_out/refpy/asan.* > /dev/null 2>&1; then head -40 $(ls $ROOT/gpurun_out/refpy/asan.* | head -1) | cut -c1-200; exit 1; fi
  echo "AddressSanitizer: no report"
fi
