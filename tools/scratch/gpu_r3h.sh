#!/bin/bash
O=gpurun_out/r3h; mkdir -p $O; export TMPDIR=/tmp PYTHONUNBUFFERED=1
( timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_reference_cuda_golden.py tests/test_gpu_fused_fullsize.py -q -m gpu --timeout 800 -x ) > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log
tail -3 $O/pytest.log
for c in cfg3 cfg2 cfg5; do
( timeout 120 python tools/kbench.py $c 30 ) 2>&1 | grep -E "KBENCH|rror" >> $O/kbench.log
done; cat $O/kbench.log
