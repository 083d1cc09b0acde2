#!/bin/bash
O=gpurun_out/r4l; mkdir -p $O; export TMPDIR=/tmp PYTHONUNBUFFERED=1
for rep in 1 2 3; do for z in "" 1; do
GHR_NO_PREZERO=$z timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-op-only 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('STEP noprezero=%s ms_per_step %.4f shard4 %.4f K8 %.4f' % ('$z' or '0', d['ms_per_step'], d['config4_shard']['ms_per_step'], d['kernels_ms']['k_render_bwd']))"
done; done
