#!/bin/bash
# round 5, same-box A/Bs: marching loss kernels with 16-B epilogue accesses (HEAD) against round 4's (variant lossold);
# K8 in k_tile_scan's heaviest-first order (GHR_TILE_ORDER bit 2)
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=$PWD/gpurun_out/r05b; mkdir -p $O; export TMPDIR=/tmp PYTHONUNBUFFERED=1
V=$PWD/build/variants
for rep in 1 2; do
  bash tools/gpu/kt.sh new$rep 2>&1 | grep -E "loss|ms_per_step|k_project\b|k_scatter|k_tile_scan" 
  bash tools/gpu/kt.sh old$rep GHR_LIB_PATH=$V/libghr_lossold.so 2>&1 | grep -E "loss|ms_per_step"
done
for s in 24 40; do bash tools/gpu/kt.sh seg$s GHR_LOSS_SEG_F=$s GHR_LOSS_SEG_B=$s 2>&1 | grep -E "loss|ms_per_step"; done
for cfg in cfg3 cfg2; do for o in 3 7 3 7; do GHR_TILE_ORDER=$o timeout 120 python tools/kbench.py $cfg 20 | sed "s/^/[order$o] /"; done; done
