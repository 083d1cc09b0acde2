#!/bin/bash
O=gpurun_out/r3u; mkdir -p $O; export TMPDIR=/tmp PYTHONUNBUFFERED=1
for v in cell scan; do
( GHR_K8=$v timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_reference_cuda_golden.py tests/test_gpu_fused_fullsize.py -q -m gpu --timeout 800 -k "not deterministic" ) > $O/pytest_$v.log 2>&1; echo "$v rc=$?"; tail -1 $O/pytest_$v.log
done
