#!/bin/bash
O=gpurun_out/r4r; mkdir -p $O; export TMPDIR=/tmp PYTHONUNBUFFERED=1
( timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu --timeout 500 -k "backward_twice" ) > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log
grep -n "Error\|passed\|failed\|^E  " $O/pytest.log | head -12
