#!/bin/bash
# round 5, call s: k_project_bwd with the slab requested behind the gather and waited for in front of the SH part
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/s; export PYTHONUNBUFFERED=1
timeout 600 python -m pytest tests/test_gpu_fused.py tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/s/tests.log 2>&1; echo "tests rc=$?"; tail -2 gpurun_out/s/tests.log
V=$PWD/build/variants
for rep in 1 2 3; do
  for v in head c1 new; do
    if [ $v = new ]; then E="GHR_NOP=1"; else E="GHR_LIB_PATH=$V/libghr_$v.so"; fi
    bash tools/gpu/kt.sh s_${v}$rep $E > gpurun_out/s/${v}$rep.txt 2>&1
    echo "$v$rep $(grep -o 'k_project[<(].*' gpurun_out/s/${v}$rep.txt | grep -o 'avg.*') | bwd $(grep -o 'k_project_bwd.*' gpurun_out/s/${v}$rep.txt | grep -o 'avg.*') | $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/s/${v}$rep.txt | head -1)"
  done
done
for v in head c1 new head c1 new; do
  if [ $v = new ]; then E="GHR_NOP=1"; else E="GHR_LIB_PATH=$V/libghr_$v.so"; fi
  env $E timeout 300 python bench.py --no-cpu-baseline --no-op-only 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v step', d['ms_per_step'], 'shard4', d['config4_shard']['ms_per_step'])"
done
