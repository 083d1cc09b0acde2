#!/bin/bash
# round 5, call za: keys per lane of the wave sort by list length
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/za; export PYTHONUNBUFFERED=1
V=$PWD/build/variants
for rep in 1 2 3; do
  for v in new sr0 sr1 sr2; do
    if [ $v = new ]; then E="GHR_NOP=1"; else E="GHR_LIB_PATH=$V/libghr_$v.so"; fi
    bash tools/gpu/kt.sh za_${v}$rep $E > gpurun_out/za/${v}$rep.txt 2>&1
    echo "$v$rep sort $(grep -o 'k_tile_sort.*' gpurun_out/za/${v}$rep.txt | grep -o 'avg.*') | $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/za/${v}$rep.txt | head -1)"
  done
done
