R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r05_final; mkdir -p $O
cd $R
( time timeout 1700 python -m pytest tests -m gpu -q -x ) > $O/pytest_gpu.log 2>&1; tail -4 $O/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE_OK')" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
timeout 300 python examples/quickstart.py > $O/quickstart.log 2>&1; echo "quickstart rc=$?"; tail -3 $O/quickstart.log
( time timeout 900 python bench.py --detail-out $O/bench_detail.json ) > $O/bench_line.json 2> $O/bench.err; tail -4 $O/bench.err; head -c 400 $O/bench_line.json
