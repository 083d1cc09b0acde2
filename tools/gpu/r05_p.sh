#!/bin/bash
# round 5, call p: k_project_bwd slab by LDS-DMA (145 VGPRs) vs registers at 168 VGPRs + 19 spilled dwords
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/p; export PYTHONUNBUFFERED=1
V=$PWD/build/variants
for rep in 1 2 3; do
  for v in head new pbwregs; do
    if [ $v = new ]; then E="GHR_NOP=1"; else E="GHR_LIB_PATH=$V/libghr_$v.so"; fi
    bash tools/gpu/kt.sh p_${v}$rep $E > gpurun_out/p/${v}$rep.txt 2>&1
    echo "$v$rep $(grep -o 'k_project(.*' gpurun_out/p/${v}$rep.txt | grep -o 'avg.*') | bwd $(grep -o 'k_project_bwd.*' gpurun_out/p/${v}$rep.txt | grep -o 'avg.*') | $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/p/${v}$rep.txt | head -1)"
  done
done
