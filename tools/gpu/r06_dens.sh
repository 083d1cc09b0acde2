#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
for i in 1 2; do
python bench.py --steps 20 --warmup 5 --no-2m --no-camera-block --no-strand-block --no-op-only --no-cpu-baseline --shard-views 0 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('ms', d['ms_per_step'], 'fixed', d['fixed_camera_step']['ms_per_step'], 'dens', d['densify_stats_step'])"
done
