#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r06sh; mkdir -p $O; export TMPDIR=/tmp
timeout 1800 python -m pytest tests/test_gpu_dist_shared.py -m gpu -x -q 2>&1 | tail -25 | tee $O/pytest_dist.log
