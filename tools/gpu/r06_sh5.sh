#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r06sh; mkdir -p $O; export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 1800 python -m pytest tests/test_gpu_dist_shared.py -m gpu -x -q 2>&1 | grep -v "amdgpu.ids\|socket.cpp\|Gloo" | tail -60 > $O/pytest_dist_full.log
