#!/bin/bash
# round 5, call x: k_project_bwd workgroup size (64 / 128 / 256 threads) x register budget (3 or 4 waves per SIMD)
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/x; export PYTHONUNBUFFERED=1
V=$PWD/build/variants
GHR_LIB_PATH=$V/libghr_b64w4.so timeout 600 python -m pytest tests/test_gpu_fused.py -x -q -m gpu > gpurun_out/x/tests.log 2>&1; echo "tests(b64w4) rc=$?"; tail -1 gpurun_out/x/tests.log
for rep in 1 2 3; do
  for v in new b64w4 b64w3 b128w4 b128w3; do
    if [ $v = new ]; then E="GHR_NOP=1"; else E="GHR_LIB_PATH=$V/libghr_$v.so"; fi
    bash tools/gpu/kt.sh x_${v}$rep $E > gpurun_out/x/${v}$rep.txt 2>&1
    echo "$v$rep bwd $(grep -o 'k_project_bwd.*' gpurun_out/x/${v}$rep.txt | grep -o 'avg.*') | $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/x/${v}$rep.txt | head -1)"
  done
done
