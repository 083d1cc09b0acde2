#!/bin/bash
O=gpurun_out/r2v; mkdir -p $O; export TMPDIR=/tmp PYTHONUNBUFFERED=1
R=$GRAFT_REPO_ROOT
V=$R/gaussianhaircut_amd/csrc/variants
export GHR_PROF_NAMES="cell acquire,pixel issue,mask list,staging,chunks,exit"
export GHR_K8=cells
for lib in $V/libghr_prof5.so; do
for c in cfg3; do
( GHR_LIB_PATH=$lib timeout 300 python tools/kbench.py $c 20 ) 2>&1 | grep -E "KBENCH|PROF|rror" >> $O/prof.log
done; done
cat $O/prof.log
