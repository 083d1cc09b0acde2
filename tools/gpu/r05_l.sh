#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 300 python -m pytest tests/test_gpu_loss_adam.py -m gpu -q -x 2>&1 | tail -2
bash tools/gpu/kt.sh l1 2>&1 | grep -E "k_loss|k_tile_scan|k_project\(|k_scatter|ms_per_step"
