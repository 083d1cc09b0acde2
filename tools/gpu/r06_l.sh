#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
P=$PWD/gpurun_out/r06l; mkdir -p $P; export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 600 python -m pytest tests/test_camera_grads.py -m gpu -q 2>&1 | tail -15 | tee $P/a.log
