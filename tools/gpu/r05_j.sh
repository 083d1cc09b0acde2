#!/bin/bash
# round 5: k_scatter with 1 / 2 / 4 Gaussians per lane, same box; parity subset first (default = 2)
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp PYTHONUNBUFFERED=1
V=$PWD/build/variants
timeout 400 python -m pytest tests/test_gpu_parity.py tests/test_reference_cuda_golden.py tests/test_gpu_reference_live.py -m gpu -q -x 2>&1 | tail -3
bash tools/gpu/kt.sh j_sc2 2>&1 | grep -E "k_scatter|ms_per_step"
bash tools/gpu/kt.sh j_sc1 GHR_LIB_PATH=$V/libghr_sc1.so 2>&1 | grep -E "k_scatter|ms_per_step"
bash tools/gpu/kt.sh j_sc4 GHR_LIB_PATH=$V/libghr_sc4.so 2>&1 | grep -E "k_scatter|ms_per_step"
bash tools/gpu/opstats.sh cfg2 | grep -E "k_scatter|fwd_ms" | cut -c1-160
