#!/bin/bash
# full -m gpu suite on the GPU box: tools/gpu/run_tests.sh [pytest args]  -> gpurun_out/tests/pytest.log
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/tests; mkdir -p $O
export TMPDIR=/tmp
T="$@"; [ -z "$T" ] && T=tests
timeout 1500 python -m pytest $T -m gpu -q 2>&1 | tail -150 > $O/pytest.log
tail -30 $O/pytest.log
