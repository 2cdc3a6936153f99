#!/bin/bash
# ThreadSanitizer run of the C++ query DAG's CPU unit tests (host/test/dag_unittest.cpp: tape store, scheduler threads,
# Dataset prefetch thread, close-while-starved): the host sources are compiled into one instrumented binary.
# gcc 11's libtsan does not intercept pthread_cond_clockwait, which is what condition_variable::wait_for compiles to: it
# then believes the mutex stays held across the wait and reports hundreds of false races and "double lock"s.  The
# instrumented copy of dag.cc therefore waits with wait_until(system_clock) (pthread_cond_timedwait, intercepted); nothing
# else differs.  Exit code 0 = no report.
set -e
R=$(cd "$(dirname "$0")/.." && pwd)/graph-learn_amd
W=${TMPDIR:-/tmp}/glx_tsan
mkdir -p $W
sed 's/\.wait_for(lock, kPoll)/.wait_until(lock, std::chrono::system_clock::now() + kPoll)/' $R/host/src/dag.cc > $W/dag_tsan.cc
g++ -std=c++17 -O1 -g -fsanitize=thread -fPIC -pthread -I$R/../include -I$R/host/include -I$R/host/test \
  $(ls $R/host/src/*.cc | grep -v "/dag.cc") $W/dag_tsan.cc $R/host/test/dag_unittest.cpp -o $W/dag_unittest_tsan \
  -L$R/lib -lglx -Wl,-rpath,$R/lib -Wl,-rpath,/opt/rocm/lib
TSAN_OPTIONS="halt_on_error=0 exitcode=66" $W/dag_unittest_tsan > $W/out.txt 2>&1 || { grep "^SUMMARY" $W/out.txt | sort | uniq -c | sort -rn | cut -c1-200 | head -20; tail -3 $W/out.txt; exit 1; }
tail -1 $W/out.txt
echo "ThreadSanitizer: no report"
