#!/bin/bash
# round 6: SH gradients as per-view factors (ABI 19): the one-process test, the two-rank tests, the projection tests
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r06sh; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_fused.py -m gpu -x -q -k "per_view_factors" 2>&1 | tail -25 | tee $O/pytest_one.log
timeout 1800 python -m pytest tests/test_gpu_dist_shared.py -m gpu -x -q 2>&1 | tail -25 | tee $O/pytest_dist.log
timeout 1800 python -m pytest tests/test_gpu_fused.py tests/test_gpu_fused_fullsize.py tests/test_camera_grads.py -m gpu -x -q 2>&1 | tail -5 | tee $O/pytest_fused.log
