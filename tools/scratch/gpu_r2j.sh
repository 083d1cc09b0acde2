#!/bin/bash
O=gpurun_out/r2j; mkdir -p $O; export TMPDIR=/tmp PYTHONUNBUFFERED=1
R=$GRAFT_REPO_ROOT
for rows in 3 0; do
  ( GHR_K8_ROWS=$rows timeout 200 python tools/kbench.py cfg3 20 ) 2>&1 | grep -E "KBENCH|rror" | sed "s/^/rows=$rows /" >> $O/kbench.log
  for so in $R/gaussianhaircut_amd/csrc/variants/libghr_*.so; do
  ( GHR_K8_ROWS=$rows GHR_LIB_PATH=$so timeout 200 python tools/kbench.py cfg3 20 ) 2>&1 | grep -E "KBENCH|PROF|rror" | sed "s/^/rows=$rows /" >> $O/kbench.log
  done
done
( GHR_K8_ROWS=3 timeout 200 python tools/kbench.py cfg2 20 ) 2>&1 | grep -E "KBENCH|rror" | sed "s/^/rows=3 /" >> $O/kbench.log
cat $O/kbench.log
