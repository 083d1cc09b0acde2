#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r06sh; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_fused.py -m gpu -x -q -k "fold_their_sh or per_view_factors or adam_fused_into" 2>&1 | tail -30 | tee $O/pytest_fold.log
