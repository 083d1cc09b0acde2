#!/bin/bash
O=gpurun_out/r2o; mkdir -p $O; export TMPDIR=/tmp PYTHONUNBUFFERED=1
R=$GRAFT_REPO_ROOT
( timeout 200 python tools/kbench.py cfg3 20 ) 2>&1 | grep -E "KBENCH|rror" >> $O/kbench.log
for so in $R/gaussianhaircut_amd/csrc/variants/libghr_b1exp*.so; do
( GHR_LIB_PATH=$so timeout 200 python tools/kbench.py cfg3 20 ) 2>&1 | grep -E "KBENCH|rror" >> $O/kbench.log
done
cat $O/kbench.log
