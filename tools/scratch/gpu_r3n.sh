#!/bin/bash
O=gpurun_out/r3n; mkdir -p $O; export TMPDIR=/tmp PYTHONUNBUFFERED=1
for rep in 1 2; do for v in cells cell; do
GHR_K8=$v timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-op-only 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('SHARD K8=$v ms_per_step %.4f shard4 %.4f' % (d['ms_per_step'], d['config4_shard']['ms_per_step']))"
done; done 2>&1 | tee $O/shard.log
