#!/bin/bash
# round 6: strand-stage direct SH gradients -- the hair tests
cd ${GRAFT_REPO_ROOT:-/root/repo}
R=$PWD; P=$PWD/gpurun_out/r06w; mkdir -p $P; export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_gpu_fused.py tests/test_reference_dropin.py tests/test_camera_grads.py tests/test_gpu_hair_fullsize.py -m gpu -x -q -k "hair or strand" 2>&1 | tail -40 | tee $P/pytest.log
