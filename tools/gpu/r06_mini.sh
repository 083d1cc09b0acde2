#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r06mini; mkdir -p $O; export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 900 python tools/mini_train.py 400 2>&1 | grep -v "amdgpu.ids" | tee $O/mini_train.txt | tail -30
