#!/bin/bash
# round 5, call u: where the tile sort zero-fills the tile's gradient lines (before the key loads / behind them / at the end)
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/u; export PYTHONUNBUFFERED=1
V=$PWD/build/variants
for rep in 1 2 3; do
  for v in new zf1 zf2; do
    if [ $v = new ]; then E="GHR_NOP=1"; else E="GHR_LIB_PATH=$V/libghr_$v.so"; fi
    bash tools/gpu/kt.sh u_${v}$rep $E > gpurun_out/u/${v}$rep.txt 2>&1
    echo "$v$rep sort $(grep -o 'k_tile_sort.*' gpurun_out/u/${v}$rep.txt | grep -o 'avg.*') | K7 $(grep -o 'k_render_fwd.*' gpurun_out/u/${v}$rep.txt | grep -o 'avg.*') | K8 $(grep -o 'k_render_bwd_cells.*' gpurun_out/u/${v}$rep.txt | grep -o 'avg.*') | $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/u/${v}$rep.txt | head -1)"
  done
done
