#!/bin/bash
# round 5, call o: k_project slab path (LDS-DMA vs registers) x raw-parameter loads (early vs late), full kernel lists
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/o; export PYTHONUNBUFFERED=1
V=$PWD/build/variants
for rep in 1 2 3; do
  for v in head new k1regs k1regslate k1dmalate; do
    if [ $v = new ]; then E="GHR_NOP=1"; else E="GHR_LIB_PATH=$V/libghr_$v.so"; fi
    bash tools/gpu/kt.sh o_${v}$rep $E > gpurun_out/o/${v}$rep.txt 2>&1
    echo "$v$rep $(grep -o 'k_project(.*' gpurun_out/o/${v}$rep.txt | grep -o 'avg.*') | bwd $(grep -o 'k_project_bwd.*' gpurun_out/o/${v}$rep.txt | grep -o 'avg.*') | $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/o/${v}$rep.txt | head -1)"
  done
done
