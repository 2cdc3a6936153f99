#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout 600 python -m pytest tests/test_gpu_dist_store.py tests/test_host_cpp.py -x -q -m gpu --timeout 600 2>&1 | tail -2
