#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_dist_shared.py -m gpu -x -q 2>&1 | grep -v "amdgpu.ids\|socket.cpp" | grep "deg \|passed\|failed" | cut -c1-900
timeout 600 python -m pytest tests/test_gpu_fused.py -m gpu -x -q -k "fold_their_sh or per_view_factors" 2>&1 | tail -3
