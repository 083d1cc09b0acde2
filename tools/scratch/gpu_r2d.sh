#!/bin/bash
O=gpurun_out/r2d; mkdir -p $O; export TMPDIR=/tmp PYTHONUNBUFFERED=1
R=$GRAFT_REPO_ROOT
( timeout 400 python tests/golden/make_reference_cuda_golden.py 2>&1 | grep -a -v "amdgpu.ids" | tail -20 ) > $O/refgolden.log 2>&1
( timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_wave_primitives.py -q -m gpu ) > $O/parity_scan.log 2>&1; echo "parity_scan rc=$?" >> $O/summary.txt
for v in scan cell; do for c in cfg3 cfg2; do
  ( GHR_K8=$v timeout 200 python tools/kbench.py $c 20 ) 2>&1 | grep -E "KBENCH|rror" >> $O/kbench.log; done; done
( timeout 1500 python -m pytest tests/test_gpu_fused_fullsize.py -q -m gpu -s ) > $O/fullsize.log 2>&1; echo "fullsize rc=$?" >> $O/summary.txt
cat $O/summary.txt; cat $O/kbench.log; cat $O/refgolden.log; tail -4 $O/parity_scan.log; grep -n "AssertionError:\|^fullsize\|passed\|failed" $O/fullsize.log
