#!/bin/bash
O=gpurun_out/r3i; mkdir -p $O; export TMPDIR=/tmp PYTHONUNBUFFERED=1
( timeout 900 python -m pytest tests/test_gpu_fused.py -q -m gpu --timeout 800 ) > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log
grep -n "AssertionError\|Error\|passed\|failed\|^FAILED\|^E  " $O/pytest.log | head -40
