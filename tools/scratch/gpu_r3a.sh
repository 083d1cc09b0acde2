#!/bin/bash
O=gpurun_out/r3a; mkdir -p $O; export TMPDIR=/tmp PYTHONUNBUFFERED=1
R=$GRAFT_REPO_ROOT
V=$R/gaussianhaircut_amd/csrc/variants
export GHR_PROF_NAMES="first acquire,acquire next,-,wait gather,chunk,exit"
export GHR_K8=cells
( timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_reference_cuda_golden.py -q -m gpu --timeout 500 -x ) > $O/pytest_cells.log 2>&1; echo "rc=$?" >> $O/pytest_cells.log
tail -4 $O/pytest_cells.log
( GHR_LIB_PATH=$V/libghr_a4.so timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_reference_cuda_golden.py -q -m gpu --timeout 500 -x ) > $O/pytest_a4.log 2>&1; echo "rc=$?" >> $O/pytest_a4.log
tail -4 $O/pytest_a4.log
for lib in "" $V/libghr_a4.so $V/libghr_noahead.so $V/libghr_a4prof.so; do
for c in cfg3 cfg2 cfg5; do
( GHR_LIB_PATH=$lib timeout 120 python tools/kbench.py $c 20 ) 2>&1 | grep -E "KBENCH|PROF|rror" >> $O/prof.log
done; done
cat $O/prof.log
