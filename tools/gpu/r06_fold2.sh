#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp PYTHONUNBUFFERED=1
O=gpurun_out/r06sh; mkdir -p $O
for rep in 1 2; do
for f in 0 1; do
GHR_FACTORED_SH_REDUCE=$f python bench.py --steps 20 --warmup 5 --no-2m --no-camera-block --no-strand-block --no-op-only --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('FOLD GHR_FACTORED_SH_REDUCE=$f headline', d['ms_per_step'], 'config4_shard', d['config4_shard']['ms_per_step'])" | tee -a $O/fold_ab.log
done
done
