#!/bin/bash
# round 5, call v: sort zero-fill last + k_project takes its atomics in before the store tail
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/v; export PYTHONUNBUFFERED=1
timeout 600 python -m pytest tests/test_gpu_fused.py tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/v/tests.log 2>&1; echo "tests rc=$?"; tail -2 gpurun_out/v/tests.log
V=$PWD/build/variants
for rep in 1 2 3; do
  for v in c2 new; do
    if [ $v = new ]; then E="GHR_NOP=1"; else E="GHR_LIB_PATH=$V/libghr_$v.so"; fi
    bash tools/gpu/kt.sh v_${v}$rep $E > gpurun_out/v/${v}$rep.txt 2>&1
    echo "$v$rep sort $(grep -o 'k_tile_sort.*' gpurun_out/v/${v}$rep.txt | grep -o 'avg.*') | K1 $(grep -o 'k_project<.*' gpurun_out/v/${v}$rep.txt | grep -o 'avg.*') | $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/v/${v}$rep.txt | head -1)"
  done
done
