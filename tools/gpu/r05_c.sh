#!/bin/bash
# round 5: K7 with the cell-mask walk on the scalar unit (variant k7swalk) -- parity subset through the variant, then same-box A/B
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=$PWD/gpurun_out/r05c; mkdir -p $O; export TMPDIR=/tmp PYTHONUNBUFFERED=1
V=$PWD/build/variants
GHR_LIB_PATH=$V/libghr_k7swalk.so timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_reference_cuda_golden.py tests/test_gpu_fused_fullsize.py -m gpu -q -x 2>&1 | tail -5
for rep in 1 2; do for cfg in cfg3 cfg2 cfg5; do
  timeout 120 python tools/kbench.py $cfg 20 | sed "s/^/[product] /"
  GHR_LIB_PATH=$V/libghr_k7swalk.so timeout 120 python tools/kbench.py $cfg 20 | sed "s/^/[k7swalk] /"
done; done
