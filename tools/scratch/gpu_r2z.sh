#!/bin/bash
O=gpurun_out/r2z; mkdir -p $O; export TMPDIR=/tmp PYTHONUNBUFFERED=1
R=$GRAFT_REPO_ROOT
V=$R/gaussianhaircut_amd/csrc/variants
export GHR_PROF_NAMES="cell acquire,gathers+pixels,mask list,wait gather,chunk,exit"
export GHR_K8=cells
for lib in "" $V/libghr_warm.so $V/libghr_profwarm.so; do
for c in cfg3 cfg2; do
( GHR_LIB_PATH=$lib timeout 120 python tools/kbench.py $c 20 ) 2>&1 | grep -E "KBENCH|PROF|rror" >> $O/prof.log
done; done
cat $O/prof.log
