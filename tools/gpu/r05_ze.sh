#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/ze; export PYTHONUNBUFFERED=1
V=$PWD/build/variants
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fused.py -x -q -m gpu > gpurun_out/ze/tests.log 2>&1; echo "tests rc=$?"; tail -2 gpurun_out/ze/tests.log
for rep in 1 2 3; do
  for v in h35 new w6 w5 nonet8; do
    if [ $v = new ]; then E="GHR_NOP=1"; else E="GHR_LIB_PATH=$V/libghr_$v.so"; fi
    bash tools/gpu/kt.sh ze_${v}$rep $E > gpurun_out/ze/${v}$rep.txt 2>&1
    echo "$v$rep sort $(grep -o "k_tile_sort.*" gpurun_out/ze/${v}$rep.txt | grep -o "avg.*") | $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/ze/${v}$rep.txt | head -1)"
  done
done
