#!/bin/bash
O=gpurun_out/r4o; mkdir -p $O; export TMPDIR=/tmp PYTHONUNBUFFERED=1
V=$GRAFT_REPO_ROOT/gaussianhaircut_amd/csrc/variants
( timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_reference_cuda_golden.py tests/test_gpu_fused_fullsize.py -q -m gpu --timeout 800 ) > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log
grep -n "AssertionError\|Error\|passed\|failed\|^FAILED\|^E  " $O/pytest.log | head -10
bash tools/stepbench.sh "" $V/libghr_pxw4.so "" $V/libghr_pxw4.so
for c in cfg3 cfg2 cfg5; do ( timeout 120 python tools/kbench.py $c 30 ) 2>&1 | grep -E "KBENCH|rror"; done
