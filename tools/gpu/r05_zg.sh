#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/zg; export PYTHONUNBUFFERED=1
V=$PWD/build/variants
GHR_LIB_PATH=$V/libghr_sc32.so timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fused.py -x -q -m gpu > gpurun_out/zg/tests.log 2>&1; echo "tests(sc32) rc=$?"; tail -2 gpurun_out/zg/tests.log
for rep in 1 2 3; do
  for v in new sc32; do
    if [ $v = new ]; then E="GHR_NOP=1"; else E="GHR_LIB_PATH=$V/libghr_$v.so"; fi
    bash tools/gpu/kt.sh zg_${v}$rep $E > gpurun_out/zg/${v}$rep.txt 2>&1
    echo "$v$rep scatter $(grep -o "k_scatter.*" gpurun_out/zg/${v}$rep.txt | grep -o "avg.*") | $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/zg/${v}$rep.txt | head -1)"
  done
done
bash tools/gpu/opstats.sh cfg2 2>&1 | grep "k_scatter\|k_tile_sort\|k_preprocess\|fwd_ms" | head -6
GHR_LIB_PATH=$V/libghr_sc32.so bash tools/gpu/opstats.sh cfg2 2>&1 | grep "k_scatter\|k_tile_sort\|k_preprocess\|fwd_ms" | head -6
