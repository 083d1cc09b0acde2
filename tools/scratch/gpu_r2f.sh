#!/bin/bash
O=gpurun_out/r2f; mkdir -p $O; export TMPDIR=/tmp PYTHONUNBUFFERED=1
R=$GRAFT_REPO_ROOT
for c in cfg3 cfg2; do
( GHR_LIB_PATH=$R/gaussianhaircut_amd/csrc/variants/libghr_prof.so timeout 200 python tools/kbench.py $c 10 ) 2>&1 | grep -E "KBENCH|PROF|rror" >> $O/kbench.log
done
( timeout 1500 python -m pytest tests/test_gpu_fused_fullsize.py -q -m gpu -s ) > $O/fullsize.log 2>&1; echo "fullsize rc=$?" >> $O/summary.txt
cat $O/summary.txt; cat $O/kbench.log; grep -n "AssertionError:\|^fullsize\|passed\|failed" $O/fullsize.log
