#!/bin/bash
O=gpurun_out/r3r; mkdir -p $O; export TMPDIR=/tmp PYTHONUNBUFFERED=1
V=$GRAFT_REPO_ROOT/gaussianhaircut_amd/csrc/variants
for rep in 1 2 3; do for lib in "" $V/libghr_sepfin.so; do
GHR_LIB_PATH=$lib timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-op-only --shard-views 0 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('STEP lib=%s ms_per_step %.4f' % ('$lib'.split('/')[-1] or 'product', d['ms_per_step']))"
done; done
