#!/bin/bash
O=gpurun_out/r4j; mkdir -p $O; export TMPDIR=/tmp PYTHONUNBUFFERED=1
( timeout 1700 python -m pytest tests -q -m gpu --timeout 1500 ) > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
grep -n "AssertionError\|Error\|passed\|failed\|^FAILED\|^E  " $O/pytest.log | head -20
for rep in 1 2 3; do
timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('STEP ms_per_step %.4f shard4 %.4f K7 %.4f K8 %.4f frac %.4f op3 fwd %.4f bwd %.4f op2 fwd %.4f bwd %.4f' % (d['ms_per_step'], d['config4_shard']['ms_per_step'], d['kernels_ms']['k_render_fwd'], d['kernels_ms']['k_render_bwd'], d['roofline']['frac'], d['op_only']['cfg3']['fwd_ms'], d['op_only']['cfg3']['bwd_ms'], d['op_only']['cfg2']['fwd_ms'], d['op_only']['cfg2']['bwd_ms']))"
done
( timeout 300 python tools/bench_hair.py 30000 100000 ) 2>&1 | grep "HAIR fused"
