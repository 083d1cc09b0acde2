#!/bin/bash
O=gpurun_out/r2g; mkdir -p $O; export TMPDIR=/tmp PYTHONUNBUFFERED=1
R=$GRAFT_REPO_ROOT
for c in cfg3; do
( GHR_LIB_PATH=$R/gaussianhaircut_amd/csrc/variants/libghr_prof.so timeout 200 python tools/kbench.py $c 10 ) 2>&1 | grep -E "KBENCH|PROF|rror" >> $O/kbench.log
done
cat $O/kbench.log
