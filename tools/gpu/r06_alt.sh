#!/bin/bash
# the alternative paths: SH gradients accumulated in place, PyTorch strand build, separate Adam pass, no workspace recycling
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r06all; mkdir -p $O; export TMPDIR=/tmp
GHR_FACTORED_SH_REDUCE=0 GHR_FUSED_STRAND_BUILD=0 GHR_FUSE_ADAM=0 GHR_RECYCLE_IMG_WS=0 timeout 2400 python -m pytest tests -m gpu -q 2>&1 | grep -v "amdgpu.ids\|socket.cpp\|Gloo" | tail -25 | tee $O/pytest_alt.log
