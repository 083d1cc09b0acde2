#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r06all; mkdir -p $O; export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "amdgpu.ids\|socket.cpp\|Gloo" | tail -40 | tee $O/pytest.log
