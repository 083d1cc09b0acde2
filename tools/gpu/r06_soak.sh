#!/bin/bash
# soak: the new paths of round 6's second half several times over on one box
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r06all; mkdir -p $O; export TMPDIR=/tmp
rm -f $O/soak.log
for i in 1 2 3 4 5; do
  timeout 900 python -m pytest tests/test_gpu_dist_shared.py tests/test_strand_build.py tests/test_gpu_fused.py tests/test_camera_grads.py -m gpu -x -q 2>&1 | tail -1 | tee -a $O/soak.log
done
