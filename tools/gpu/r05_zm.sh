#!/bin/bash
# round 5, call zm: the tile sort leaves a small rect's gradient-line positions in the Gaussian's own row; the per-Gaussian
# gather has them in its first round trip
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/zm; export PYTHONUNBUFFERED=1
V=$PWD/build/variants
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fused.py -x -q -m gpu > gpurun_out/zm/tests.log 2>&1; echo "tests rc=$?"; tail -2 gpurun_out/zm/tests.log
for rep in 1 2 3; do
  for v in c7 new; do
    if [ $v = new ]; then E="GHR_NOP=1"; else E="GHR_LIB_PATH=$V/libghr_$v.so"; fi
    bash tools/gpu/kt.sh zm_${v}$rep $E > gpurun_out/zm/${v}$rep.txt 2>&1
    echo "$v$rep bwd $(grep -o "k_project_bwd.*" gpurun_out/zm/${v}$rep.txt | grep -o "avg.*") | sort $(grep -o "k_tile_sort.*" gpurun_out/zm/${v}$rep.txt | grep -o "avg.*") | $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/zm/${v}$rep.txt | head -1)"
  done
done
