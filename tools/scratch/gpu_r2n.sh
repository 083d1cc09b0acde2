#!/bin/bash
O=gpurun_out/r2n; mkdir -p $O; export TMPDIR=/tmp PYTHONUNBUFFERED=1
for v in cell scan; do for c in cfg3 cfg2 cfg5; do
  ( GHR_K8=$v timeout 200 python tools/kbench.py $c 20 ) 2>&1 | grep -E "KBENCH|rror" | sed "s/^/$v /" >> $O/kbench.log; done; done
( timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_reference_cuda_golden.py -q -m gpu ) > $O/parity_cell.log 2>&1; echo "parity_cell rc=$?" >> $O/summary.txt
cat $O/summary.txt; cat $O/kbench.log; tail -3 $O/parity_cell.log
