#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/zd; export PYTHONUNBUFFERED=1
V=$PWD/build/variants
for rep in 1 2; do
  for v in new nonet; do
    if [ $v = new ]; then E="GHR_NOP=1"; else E="GHR_LIB_PATH=$V/libghr_$v.so"; fi
    bash tools/gpu/kt.sh zd_${v}$rep $E > gpurun_out/zd/${v}$rep.txt 2>&1
    echo "$v$rep sort $(grep -o 'k_tile_sort.*' gpurun_out/zd/${v}$rep.txt | grep -o 'avg.*') | $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/zd/${v}$rep.txt | head -1)"
  done
done
