#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
P=$PWD/gpurun_out/r06k; mkdir -p $P; export TMPDIR=/tmp PYTHONUNBUFFERED=1
F="--no-op-only --no-cpu-baseline --no-2m --no-camera-block --shard-views 0"
for args in "--steps 20 --warmup 5" "--steps 20 --warmup 5" "--steps 100 --warmup 5" "--steps 20 --warmup 5 --cameras 1" "--steps 20 --warmup 40"; do
  python bench.py $F $args 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('BENCH [$args] ms_per_step', d['ms_per_step'], 'fixed', d.get('fixed_camera_step'), 'dens', d.get('densify_stats_step'))" | tee -a $P/bench_variants.log
done
