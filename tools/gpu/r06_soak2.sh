#!/bin/bash
# final soak of round 6: the end-to-end stage-1 loop (densify events grow the lists into the dense-tile kernels), 100 strand iterations,
# and the tests of this session's paths three times over, on one box
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r06soak2; mkdir -p $O; export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 900 python tools/mini_train.py 400 2>&1 | grep -v "amdgpu.ids" | tee $O/mini_train.txt | tail -14
timeout 600 python tools/strandstep.py 100 2>&1 | grep STRAND | tee $O/strand_100.txt
rm -f $O/soak.log
for i in 1 2 3; do
  timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_hair_fullsize.py tests/test_strand_build.py tests/test_gpu_loss_adam.py tests/test_gpu_fused.py tests/test_gpu_dist_shared.py -m gpu -x -q 2>&1 | tail -1 | tee -a $O/soak.log
done
