#!/bin/bash
# round 5: k_tile_sort4 (four tiles per workgroup, one wave each) against k_tile_sort, same box; parity subset first (sort4 default)
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=$PWD/gpurun_out/r05i; mkdir -p $O; export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 400 python -m pytest tests/test_gpu_parity.py tests/test_reference_cuda_golden.py tests/test_gpu_reference_live.py -m gpu -q -x 2>&1 | tail -4
bash tools/gpu/kt.sh i_sort4 2>&1 | grep -E "k_tile_sort|k_render_fwd|ms_per_step"
bash tools/gpu/kt.sh i_sort1 GHR_SORT4=0 2>&1 | grep -E "k_tile_sort|k_render_fwd|ms_per_step"
bash tools/gpu/opstats.sh cfg2 | grep -E "k_tile_sort|fwd_ms" | cut -c1-200
