# Round-6 final check at the round's last sources (one gpurun call): the full GPU suite, smoke, quickstart, the driver's own
# bench command (line + detail), then the three P = 8 rigs of DESIGN 12.   gpurun --timeout 3000 -- bash scripts/r06/final_check.sh [tag]
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=${1:-run1}; O=$R/gpurun_out/r06_final; mkdir -p $O
cd $R
( time timeout 1700 python -m pytest tests -m gpu -q -x ) > $O/pytest_gpu.log 2>&1; tail -4 $O/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE_OK')" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
timeout 300 python examples/quickstart.py > $O/quickstart.log 2>&1; echo "quickstart rc=$?"; tail -3 $O/quickstart.log
( time timeout 900 python bench.py --detail-out $O/bench_c3_n1_${TAG}_detail.json ) > $O/bench_c3_n1_${TAG}_line.json 2> $O/bench_${TAG}.err; tail -4 $O/bench_${TAG}.err; head -c 300 $O/bench_c3_n1_${TAG}_line.json; echo
if [ -z "$SKIP_RIGS" ]; then
  rm -f $R/gpurun_out/r03/p8.txt
  bash scripts/r06/gpu.sh p8-sym p8-solo p8 2>&1 | grep -E "engine kernels|per rank-step|ONLY rank 0" | cut -c1-260
  cp $R/gpurun_out/r06/p8_sym_step.txt $R/gpurun_out/r06/p8_solo_step.txt $R/gpurun_out/r06/p8_sym_run.txt $R/gpurun_out/r06/p8_all_asking.txt $O/ 2>/dev/null
fi
