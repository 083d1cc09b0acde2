#!/bin/bash
O=gpurun_out/r2i; mkdir -p $O; export TMPDIR=/tmp PYTHONUNBUFFERED=1
R=$GRAFT_REPO_ROOT
( timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_reference_cuda_golden.py -q -m gpu -x ) > $O/parity_rows.log 2>&1; echo "parity_rows rc=$?" >> $O/summary.txt
for rows in 3 0; do for c in cfg3 cfg2; do
  ( GHR_K8_ROWS=$rows timeout 200 python tools/kbench.py $c 20 ) 2>&1 | grep -E "KBENCH|rror" | sed "s/^/rows=$rows /" >> $O/kbench.log; done; done
( GHR_K8_ROWS=6 timeout 200 python tools/kbench.py cfg2 20 ) 2>&1 | grep -E "KBENCH|rror" | sed "s/^/rows=6 /" >> $O/kbench.log
( timeout 1500 python -m pytest tests/test_gpu_fused_fullsize.py tests/test_gpu_fused.py -q -m gpu -s ) > $O/fullsize.log 2>&1; echo "fullsize rc=$?" >> $O/summary.txt
cat $O/summary.txt; cat $O/kbench.log; tail -5 $O/parity_rows.log; grep -n "AssertionError:\|^fullsize\|passed\|failed" $O/fullsize.log
