#!/bin/bash
# round 6: fused strand build (ABI 18) -- its tests, the hair tests, the iteration's breakdown
cd ${GRAFT_REPO_ROOT:-/root/repo}
R=$PWD; P=$PWD/gpurun_out/r06x; mkdir -p $P; export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_strand_build.py -m gpu -x -q 2>&1 | tail -30 | tee $P/pytest_build.log
timeout 900 python -m pytest tests/test_gpu_fused.py tests/test_reference_dropin.py tests/test_camera_grads.py tests/test_gpu_hair_fullsize.py -m gpu -x -q -k "hair or strand" 2>&1 | tail -30 | tee $P/pytest.log
GHR_FUSED_STRAND_BUILD=0 python tools/strandstep.py 20 2>&1 | grep STRAND | sed 's/^/torch-build /' | tee $P/strand_ab.log
python tools/strandstep.py 20 2>&1 | grep STRAND | sed 's/^/hip-build   /' | tee -a $P/strand_ab.log
bash tools/gpu/r06_v.sh
cp -r gpurun_out/r06v/* $P/ 2>/dev/null
