#!/bin/bash
# final-tree validation: the whole -m gpu suite, then the two fast-path knobs switched off
cd ${GRAFT_REPO_ROOT:-/root/repo}
P=$PWD/gpurun_out/r06t; mkdir -p $P; export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -4 | tee $P/pytest_all.log
GHR_FUSE_ADAM=0 GHR_RECYCLE_IMG_WS=0 timeout 900 python -m pytest tests/test_gpu_fused.py tests/test_gpu_parity.py tests/test_camera_grads.py -m gpu -q 2>&1 | tail -4 | tee $P/pytest_knobs_off.log
