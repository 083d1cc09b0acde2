#!/bin/bash
# round 5 probes: k_scatter without its atomics (timing only), K1's counting atomics returning (timing only; results unchanged)
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp PYTHONUNBUFFERED=1
V=$PWD/build/variants
for rep in 1 2; do
  bash tools/gpu/kt.sh g_head$rep 2>&1 | grep -E "k_project\(|k_scatter|ms_per_step"
  bash tools/gpu/kt.sh g_noatom$rep GHR_LIB_PATH=$V/libghr_scnoatom.so 2>&1 | grep -E "k_project\(|k_scatter|ms_per_step"
  bash tools/gpu/kt.sh g_k1ret$rep GHR_LIB_PATH=$V/libghr_k1ret.so 2>&1 | grep -E "k_project\(|k_scatter|ms_per_step"
done
