#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 300 ./graph-learn_amd/lib/partition_stitch_unittest 2>&1 | tail -15
