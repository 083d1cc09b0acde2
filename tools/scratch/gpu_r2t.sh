#!/bin/bash
O=gpurun_out/r2t; mkdir -p $O; export TMPDIR=/tmp PYTHONUNBUFFERED=1
R=$GRAFT_REPO_ROOT
V=$R/gaussianhaircut_amd/csrc/variants
export GHR_PROF_NAMES="zero+barrier,cell fetch/pixels/idle,mask list,staging,chunks,-"
export GHR_K8=cells
for lib in $V/libghr_w5.so $V/libghr_w5na.so $V/libghr_prof5.so $V/libghr_prof5na.so; do
for c in cfg3; do
( GHR_LIB_PATH=$lib timeout 300 python tools/kbench.py $c 20 ) 2>&1 | grep -E "KBENCH|PROF|rror" >> $O/prof.log
done; done
cat $O/prof.log
