#!/bin/bash
# round 5: list positions handed out by K1's counting atomics (k_scatter without atomics for small rects) against the previous
# binning (variant prevbin), same box; parity subset first
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp PYTHONUNBUFFERED=1
V=$PWD/build/variants
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_reference_cuda_golden.py tests/test_gpu_reference_live.py tests/test_gpu_fused.py tests/test_gpu_hair_fullsize.py -m gpu -q -x 2>&1 | tail -3
for rep in 1 2; do
bash tools/gpu/kt.sh k_new$rep 2>&1 | grep -E "k_scatter|k_project\(|k_tile_scan|k_tile_sort|ms_per_step"
bash tools/gpu/kt.sh k_prev$rep GHR_LIB_PATH=$V/libghr_prevbin.so 2>&1 | grep -E "k_scatter|k_project\(|k_tile_scan|k_tile_sort|ms_per_step"
done
bash tools/gpu/opstats.sh cfg2 | grep -E "k_scatter|k_preprocess|k_tile_scan|fwd_ms" | cut -c1-160
