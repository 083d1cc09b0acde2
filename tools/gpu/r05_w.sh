#!/bin/bash
# round 5, call w: tile + list span of a workgroup in one load (sort, K7, K8) and k_scatter's loads branch-free / store last
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/w; export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_gpu_fused.py tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/w/tests.log 2>&1; echo "tests rc=$?"; tail -2 gpurun_out/w/tests.log
V=$PWD/build/variants
for rep in 1 2 3; do
  for v in c3 new; do
    if [ $v = new ]; then E="GHR_NOP=1"; else E="GHR_LIB_PATH=$V/libghr_$v.so"; fi
    bash tools/gpu/kt.sh w_${v}$rep $E > gpurun_out/w/${v}$rep.txt 2>&1
    echo "$v$rep sort $(grep -o 'k_tile_sort.*' gpurun_out/w/${v}$rep.txt | grep -o 'avg.*') | K7 $(grep -o 'k_render_fwd.*' gpurun_out/w/${v}$rep.txt | grep -o 'avg.*') | K8 $(grep -o 'k_render_bwd_cells.*' gpurun_out/w/${v}$rep.txt | grep -o 'avg.*') | scat $(grep -o 'k_scatter.*' gpurun_out/w/${v}$rep.txt | grep -o 'avg.*') | scan $(grep -o 'k_tile_scan.*' gpurun_out/w/${v}$rep.txt | grep -o 'avg.*') | $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/w/${v}$rep.txt | head -1)"
  done
done
GHR_TILE_ORDER=7 bash tools/gpu/kt.sh w_new_o7 > gpurun_out/w/new_o7.txt 2>&1; echo "order7 K8 $(grep -o 'k_render_bwd_cells.*' gpurun_out/w/new_o7.txt | grep -o 'avg.*') $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/w/new_o7.txt | head -1)"
