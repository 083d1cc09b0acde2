#!/bin/bash
# soak: the backward parity tests of the cell-list K8 over and over (intermittent ordering bugs would show here)
O=gpurun_out/r4c; mkdir -p $O; export TMPDIR=/tmp PYTHONUNBUFFERED=1
fail=0
for i in $(seq 1 12); do
  timeout 300 python -m pytest tests/test_gpu_parity.py -q -m gpu -x --timeout 250 -k "cfg2 or cfg3 or long_tile or vs_oracle" > $O/run_$i.log 2>&1 || { fail=$((fail+1)); tail -5 $O/run_$i.log; }
done
echo "soak: 12 rounds, $fail failed"; tail -1 $O/run_12.log
