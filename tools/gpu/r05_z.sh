#!/bin/bash
# round 5, call z: tile sort as one wave per tile with the register-blocked network
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/z; export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fused.py -x -q -m gpu > gpurun_out/z/tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/z/tests.log
V=$PWD/build/variants
for rep in 1 2 3; do
  for v in c4 new; do
    if [ $v = new ]; then E="GHR_NOP=1"; else E="GHR_LIB_PATH=$V/libghr_$v.so"; fi
    bash tools/gpu/kt.sh z_${v}$rep $E > gpurun_out/z/${v}$rep.txt 2>&1
    echo "$v$rep sort $(grep -o 'k_tile_sort.*' gpurun_out/z/${v}$rep.txt | grep -o 'avg.*') | K7 $(grep -o 'k_render_fwd.*' gpurun_out/z/${v}$rep.txt | grep -o 'avg.*') | $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/z/${v}$rep.txt | head -1)"
  done
done
