#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/zh; export PYTHONUNBUFFERED=1
V=$PWD/build/variants
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fused.py -x -q -m gpu > gpurun_out/zh/tests.log 2>&1; echo "tests rc=$?"; tail -2 gpurun_out/zh/tests.log
for rep in 1 2 3; do
  for v in c5 new; do
    if [ $v = new ]; then E="GHR_NOP=1"; else E="GHR_LIB_PATH=$V/libghr_$v.so"; fi
    bash tools/gpu/kt.sh zh_${v}$rep $E > gpurun_out/zh/${v}$rep.txt 2>&1
    echo "$v$rep sort $(grep -o "k_tile_sort.*" gpurun_out/zh/${v}$rep.txt | grep -o "avg.*") | scan $(grep -o "k_tile_scan.*" gpurun_out/zh/${v}$rep.txt | grep -o "avg.*") | K7 $(grep -o "k_render_fwd.*" gpurun_out/zh/${v}$rep.txt | grep -o "avg.*") | $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/zh/${v}$rep.txt | head -1)"
  done
done
