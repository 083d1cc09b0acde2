#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/zj; export PYTHONUNBUFFERED=1
V=$PWD/build/variants
for rep in 1 2 3; do
  for v in new slabnt; do
    if [ $v = new ]; then E="GHR_NOP=1"; else E="GHR_LIB_PATH=$V/libghr_$v.so"; fi
    bash tools/gpu/kt.sh zj_${v}$rep $E > gpurun_out/zj/${v}$rep.txt 2>&1
    echo "$v$rep bwd $(grep -o "k_project_bwd.*" gpurun_out/zj/${v}$rep.txt | grep -o "avg.*") | adam $(grep -o "k_adam_v4.*" gpurun_out/zj/${v}$rep.txt | grep -o "avg.*") | $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/zj/${v}$rep.txt | head -1)"
  done
done
