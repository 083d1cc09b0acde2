#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r06sh; mkdir -p $O; export TMPDIR=/tmp PYTHONUNBUFFERED=1
SHARDSTEP_PROFILE=1 MASTER_PORT=29584 python tools/shardstep.py 1 30 2>&1 | grep -v "amdgpu.ids\|socket.cpp" | tee $O/shardstep_host_profile.txt | head -90
