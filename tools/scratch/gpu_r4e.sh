#!/bin/bash
O=gpurun_out/r4e; mkdir -p $O; export TMPDIR=/tmp PYTHONUNBUFFERED=1
( timeout 1700 python -m pytest tests -q -m gpu --timeout 1500 ) > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
grep -n "AssertionError\|Error\|passed\|failed\|^FAILED\|^E  " $O/pytest.log | head -20
for v in cell scan; do
( GHR_K8=$v timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fused_fullsize.py -q -m gpu --timeout 500 -k "not deterministic" ) > $O/pytest_$v.log 2>&1; echo "$v rc=$?"; tail -1 $O/pytest_$v.log
done
for c in cfg3 cfg2 cfg5; do ( timeout 120 python tools/kbench.py $c 30 ) 2>&1 | grep -E "KBENCH|rror"; done
bash tools/stepbench.sh "" ""
( timeout 300 python tools/bench_hair.py 30000 100000 ) 2>&1 | grep "HAIR fused"
