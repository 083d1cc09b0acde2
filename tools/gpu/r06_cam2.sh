#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
R=$PWD; O=$PWD/gpurun_out/r06cam; mkdir -p $O; export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_camera_grads.py tests/test_gpu_fused.py -m gpu -x -q 2>&1 | tail -3
for rep in 1 2; do for mode in plain leaf; do python tools/camstep.py $mode 60 2>&1 | grep CAMSTEP | tee -a $O/camstep_no_scratch.log; done; done
