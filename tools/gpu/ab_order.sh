#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r3k; mkdir -p $O; rm -f $O/*
export TMPDIR=/tmp
for ord in 0 1 2 3 7 0 3; do
GHR_TILE_ORDER=$ord timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-op-only 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[order mask $ord] step', d['ms_per_step'], d['kernels_ms']['k_render_fwd'], d['kernels_ms']['k_render_bwd'], 'shard', d['config4_shard']['ms_per_step'])" >> $O/kbench.log
done
cat $O/kbench.log
