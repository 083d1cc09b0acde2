#!/bin/bash
O=gpurun_out/r3q; mkdir -p $O; export TMPDIR=/tmp PYTHONUNBUFFERED=1
( timeout 900 python -m pytest tests/test_gpu_fused.py tests/test_gpu_dist_shared.py tests/test_gpu_loss_adam.py tests/test_gpu_fused_fullsize.py -q -m gpu --timeout 800 ) > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log
grep -n "AssertionError\|Error\|passed\|failed\|^FAILED\|^E  " $O/pytest.log | head -20
for rep in 1 2; do
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-op-only 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('STEP ms_per_step %.4f shard4 %.4f' % (d['ms_per_step'], d['config4_shard']['ms_per_step']))"
done
