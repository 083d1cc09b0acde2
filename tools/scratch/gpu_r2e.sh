#!/bin/bash
O=gpurun_out/r2e; mkdir -p $O; export TMPDIR=/tmp PYTHONUNBUFFERED=1
R=$GRAFT_REPO_ROOT
( GHR_LIB_PATH=$R/gaussianhaircut_amd/csrc/variants/libghr_prof.so timeout 200 python tools/kbench.py cfg3 10 ) 2>&1 | grep -E "KBENCH|PROF|rror" >> $O/kbench.log
( GHR_LIB_PATH=$R/gaussianhaircut_amd/csrc/variants/libghr_prof.so timeout 200 python tools/kbench.py cfg2 10 ) 2>&1 | grep -E "KBENCH|PROF|rror" >> $O/kbench.log
( timeout 600 python -m pytest tests/test_reference_cuda_golden.py -q -m gpu ) > $O/refgold_gpu.log 2>&1; echo "refgold_gpu rc=$?" >> $O/summary.txt
( timeout 1500 python -m pytest tests/test_gpu_fused_fullsize.py -q -m gpu -s ) > $O/fullsize.log 2>&1; echo "fullsize rc=$?" >> $O/summary.txt
cat $O/summary.txt; cat $O/kbench.log; tail -4 $O/refgold_gpu.log; grep -n "AssertionError:\|^fullsize\|passed\|failed" $O/fullsize.log
