#!/bin/bash
# round 5, call t: tile sort with the rect gather requested before the network (keys in registers, binary search afterwards)
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/t; export PYTHONUNBUFFERED=1
timeout 600 python -m pytest tests/test_gpu_fused.py tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/t/tests.log 2>&1; echo "tests rc=$?"; tail -2 gpurun_out/t/tests.log
V=$PWD/build/variants
for rep in 1 2 3; do
  for v in c2 new; do
    if [ $v = new ]; then E="GHR_NOP=1"; else E="GHR_LIB_PATH=$V/libghr_$v.so"; fi
    bash tools/gpu/kt.sh t_${v}$rep $E > gpurun_out/t/${v}$rep.txt 2>&1
    echo "$v$rep sort $(grep -o 'k_tile_sort.*' gpurun_out/t/${v}$rep.txt | grep -o 'avg.*') | K7 $(grep -o 'k_render_fwd.*' gpurun_out/t/${v}$rep.txt | grep -o 'avg.*') | $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/t/${v}$rep.txt | head -1)"
  done
done
