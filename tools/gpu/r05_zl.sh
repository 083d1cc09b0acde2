#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/zl; export PYTHONUNBUFFERED=1
V=$PWD/build/variants
timeout 900 python -m pytest tests/test_gpu_loss_adam.py -x -q -m gpu 2>&1 | tail -2
for rep in 1 2 3; do
  for v in c6 new; do
    if [ $v = new ]; then E="GHR_NOP=1"; else E="GHR_LIB_PATH=$V/libghr_$v.so"; fi
    bash tools/gpu/kt.sh zl_${v}$rep $E > gpurun_out/zl/${v}$rep.txt 2>&1
    echo "$v$rep lossfwd $(grep -o "k_loss_fwd_cached_v.*" gpurun_out/zl/${v}$rep.txt | grep -o "avg.*") | lossbwd $(grep -o "k_loss_bwd_v.*" gpurun_out/zl/${v}$rep.txt | grep -o "avg.*") | $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/zl/${v}$rep.txt | head -1)"
  done
done
