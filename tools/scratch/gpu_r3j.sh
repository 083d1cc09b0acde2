#!/bin/bash
O=gpurun_out/r3j; mkdir -p $O; export TMPDIR=/tmp PYTHONUNBUFFERED=1
( timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu --timeout 800 -k "deterministic or cfg1 or tiny" ) > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log
grep -n "AssertionError\|Error\|passed\|failed\|^FAILED\|^E  " $O/pytest.log | head -20
