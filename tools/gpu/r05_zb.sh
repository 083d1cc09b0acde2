#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/zb; rm -f gpurun_out/zb/sq.txt
bash tools/gpu/sqpmc_sort.sh gpurun_out/zb/sq.txt cfg3
bash tools/gpu/sqpmc_sort.sh gpurun_out/zb/sq.txt cfg3 GHR_LIB_PATH=$PWD/build/variants/libghr_c4.so
cat gpurun_out/zb/sq.txt
