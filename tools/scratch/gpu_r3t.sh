#!/bin/bash
O=gpurun_out/r3t; mkdir -p $O; export TMPDIR=/tmp PYTHONUNBUFFERED=1
V=$GRAFT_REPO_ROOT/gaussianhaircut_amd/csrc/variants
export GHR_PROF_NAMES="loop top,barrier wait,staging+cull params,cell masks,pass loop,epilogue"
for c in cfg3 cfg2; do
( GHR_PROF_K7=1 GHR_LIB_PATH=$V/libghr_k7prof.so timeout 120 python tools/kbench.py $c 10 ) 2>&1 | grep -E "KBENCH|PROF|rror"
done
