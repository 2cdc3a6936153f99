#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=$PWD/gpurun_out/r02_run39
mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -x -q -m gpu --timeout 900 > $O/pytest_all.log 2>&1
echo "pytest_all rc=$?"
grep -n "passed\|failed" $O/pytest_all.log | tail -2
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
