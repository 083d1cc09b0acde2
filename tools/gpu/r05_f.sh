#!/bin/bash
# round 5: big rects expanded load-balanced over the workgroup (tile counting + scatter): parity subset, then the op's kernel breakdown
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_reference_cuda_golden.py tests/test_gpu_fused_fullsize.py tests/test_gpu_reference_live.py tests/test_gpu_fused.py -m gpu -q -x 2>&1 | tail -6
bash tools/gpu/opstats.sh cfg2
bash tools/gpu/opstats.sh cfg3
bash tools/gpu/kt.sh f1 2>&1 | grep -E "k_project\(|k_scatter|k_tile_scan|k_tile_sort|ms_per_step|fillBuffer"
