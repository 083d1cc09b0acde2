#!/bin/bash
# round 5, call r: k_project with the geometry + counting atomics under the slab's round trip (branch-free loads)
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r; export PYTHONUNBUFFERED=1
timeout 600 python -m pytest tests/test_gpu_fused.py tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/r/tests.log 2>&1; echo "tests rc=$?"; tail -2 gpurun_out/r/tests.log
V=$PWD/build/variants
for rep in 1 2 3; do
  for v in head new; do
    if [ $v = new ]; then E="GHR_NOP=1"; else E="GHR_LIB_PATH=$V/libghr_$v.so"; fi
    bash tools/gpu/kt.sh r_${v}$rep $E > gpurun_out/r/${v}$rep.txt 2>&1
    echo "$v$rep $(grep -o 'k_project[<(].*' gpurun_out/r/${v}$rep.txt | grep -o 'avg.*') | bwd $(grep -o 'k_project_bwd.*' gpurun_out/r/${v}$rep.txt | grep -o 'avg.*') | $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/r/${v}$rep.txt | head -1)"
  done
done
