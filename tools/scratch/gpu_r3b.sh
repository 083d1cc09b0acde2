#!/bin/bash
O=gpurun_out/r3b; mkdir -p $O; export TMPDIR=/tmp PYTHONUNBUFFERED=1
R=$GRAFT_REPO_ROOT
V=$R/gaussianhaircut_amd/csrc/variants
for rep in 1 2; do
for lib in "" $V/libghr_k7nomask.so $V/libghr_k7nolast.so $V/libghr_k7none.so; do
( GHR_LIB_PATH=$lib timeout 120 python tools/kbench.py cfg3 30 ) 2>&1 | grep -E "KBENCH|rror" >> $O/k7.log
done; done
cat $O/k7.log
