#!/bin/bash
# K8 "cells" variant: parity + kbench vs cell kernel
O=gpurun_out/r2q; mkdir -p $O; export TMPDIR=/tmp PYTHONUNBUFFERED=1
for v in cell cells; do
for c in cfg3 cfg2 cfg5; do
( GHR_K8=$v timeout 300 python tools/kbench.py $c 20 ) 2>&1 | grep -E "KBENCH|rror" | sed "s/^/$v /" >> $O/kbench.log
done; done
cat $O/kbench.log
( GHR_K8=cells timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_reference_cuda_golden.py tests/test_gpu_fused_fullsize.py tests/test_gpu_fused.py -q -m gpu --timeout 1000 -x ) > $O/pytest_cells.log 2>&1; echo "rc=$?" >> $O/pytest_cells.log
tail -15 $O/pytest_cells.log
