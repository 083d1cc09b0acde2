#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/zi; export PYTHONUNBUFFERED=1
V=$PWD/build/variants
GHR_LIB_PATH=$V/libghr_adamearly.so timeout 600 python -m pytest tests/test_gpu_loss_adam.py -x -q -m gpu -k "adam or Adam" 2>&1 | tail -2
for rep in 1 2 3; do
  for v in new adamearly; do
    if [ $v = new ]; then E="GHR_NOP=1"; else E="GHR_LIB_PATH=$V/libghr_$v.so"; fi
    bash tools/gpu/kt.sh zi_${v}$rep $E > gpurun_out/zi/${v}$rep.txt 2>&1
    echo "$v$rep adam $(grep -o "k_adam_v4.*" gpurun_out/zi/${v}$rep.txt | grep -o "avg.*") | $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/zi/${v}$rep.txt | head -1)"
  done
done
