#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp PYTHONUNBUFFERED=1
V=$PWD/build/variants
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_reference_cuda_golden.py -m gpu -q -x 2>&1 | tail -2
for rep in 1 2; do
bash tools/gpu/kt.sh k_new$rep 2>&1 | grep -E "k_scatter|k_project\(|ms_per_step"
bash tools/gpu/kt.sh k_prev$rep GHR_LIB_PATH=$V/libghr_prevbin.so 2>&1 | grep -E "k_scatter|k_project\(|ms_per_step"
done
