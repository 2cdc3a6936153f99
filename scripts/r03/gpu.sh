#!/bin/bash
# Round-3 GPU leases, one entry point:   gpurun --timeout N -- bash scripts/r03/gpu.sh <what> [pytest args]
#   tests [args]   pytest -m gpu (default: the whole suite) -> gpurun_out/r03/pytest.log
#   all            tests, then scripts/r03/profile.sh (default bench line + rocprofv3 traces / PMC passes per workload);
#                  afterwards: python scripts/r03/summarize.py   (writes profiles/r03/SUMMARY.md, profiles/pmc_traffic.json)
#   world1         bench.py's N > 1 code path with ONE rank over RCCL, every leg verified
#   p8             eight ranks as threads on one GPU (scripts/edge_cut_p8_probe.py): count exchange per request /
#                  merged aggregation / speculation ledger
#   p8-solo        the same with only rank 0 asking, under rocprofv3: per-kernel cost of one rank's step + the owners' service
#   p8-sym         every rank asking on ONE stream, under rocprofv3, rank 0's launches only: one rank's step in the
#                  symmetric case (its own request + one launch per hop / one row gather for all seven peers)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03; mkdir -p $O
cd $R
what=$1; shift
case "$what" in
  tests)
    timeout 1500 python -m pytest ${@:-tests} -m gpu -q -x > $O/pytest.log 2>&1; tail -6 $O/pytest.log ;;
  all)
    timeout 1500 python -m pytest tests -m gpu -q -x > $O/pytest_all.log 2>&1; tail -6 $O/pytest_all.log
    bash scripts/r03/profile.sh 2>&1 | tail -30 ;;
  world1)
    GLX_DIST_NO_SHORTCUT=1 timeout 600 python bench.py --gpus 1 --force-sharded --cpu-baseline off --host-boundary off \
      --roofline-probes off --edge-cut-probe off --small-batches off --other-configs "" > $O/bench_world1.json 2> $O/bench_world1.err
    tail -3 $O/bench_world1.err ;;
  p8)
    for mode in "0 0" "0 1" "1 1"; do
      set -- $mode
      echo "=== LEDGER=$1 MERGED=$2" >> $O/p8.txt
      GRAPH_REPLICA=1 LEDGER=$1 MERGED=$2 timeout 600 python scripts/edge_cut_p8_probe.py 8 ${HOT:-0.25} 6 2>&1 | grep -v amdgpu.ids >> $O/p8.txt
    done
    cut -c1-300 $O/p8.txt ;;
  p8-solo)
    cd /tmp && export TMPDIR=/tmp
    GRAPH_REPLICA=1 MERGED=1 timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof_solo -o p8 --output-format csv -- \
      python $R/scripts/edge_cut_p8_probe.py 8 ${HOT:-0.25} 6 solo 2>&1 | grep "ONLY rank 0"
    python $R/scripts/r03/p8_solo_step.py $(find $O/prof_solo -name '*kernel_trace.csv' | head -1) | tee $O/p8_solo_step.txt
    rm -rf $O/prof_solo ;;
  p8-sym)
    # every rank asks, one stream for all (no kernel overlaps another): rank 0's host thread's kernels = ONE rank's step
    cd /tmp && export TMPDIR=/tmp
    GRAPH_REPLICA=1 MERGED=1 LEDGER=${LEDGER:-0} timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof_sym -o p8 --output-format csv -- \
      python $R/scripts/edge_cut_p8_probe.py 8 ${HOT:-0.25} 6 sym 2>&1 | grep -E "every rank asks|rank 0 host thread" | tee $O/p8_sym_run.txt
    tid=$(grep "rank 0 host thread" $O/p8_sym_run.txt | grep -o '[0-9]*$')
    python $R/scripts/r03/p8_solo_step.py $(find $O/prof_sym -name '*kernel_trace.csv' | head -1) --thread $tid | tee $O/p8_sym_step.txt
    rm -rf $O/prof_sym ;;
  *) echo "usage: gpu.sh tests|all|world1|p8|p8-solo|p8-sym"; exit 2 ;;
esac
