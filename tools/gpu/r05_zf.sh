#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/zf; export PYTHONUNBUFFERED=1
for rep in 1 2 3; do
  for v in order3 order2; do
    if [ $v = order3 ]; then E="GHR_TILE_ORDER=3"; else E="GHR_TILE_ORDER=2"; fi
    bash tools/gpu/kt.sh zf_${v}$rep $E > gpurun_out/zf/${v}$rep.txt 2>&1
    echo "$v$rep sort $(grep -o "k_tile_sort.*" gpurun_out/zf/${v}$rep.txt | grep -o "avg.*") | K7 $(grep -o "k_render_fwd.*" gpurun_out/zf/${v}$rep.txt | grep -o "avg.*") | $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/zf/${v}$rep.txt | head -1)"
  done
done
