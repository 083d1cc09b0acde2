#!/bin/bash
# round 6: do full garbage collections fall into bench.py's timed regions, and is that what the sporadic 2.8-3.7 ms side blocks are?
cd ${GRAFT_REPO_ROOT:-/root/repo}
R=$PWD; O=$PWD/gpurun_out/r06gc; mkdir -p $O; export TMPDIR=/tmp PYTHONUNBUFFERED=1
rm -f $O/gc.log
B="python $R/bench.py --no-cpu-baseline --no-op-only"
for rep in 1 2 3 4 5 6; do
for nf in 1 0; do
GHR_BENCH_NO_GC_FREEZE=$nf $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[no_freeze=$nf] headline', d['ms_per_step'], 'fixed', d['fixed_camera_step']['ms_per_step'], 'config5_2M', d['config5_2M']['ms_per_step'], 'shard', d['config4_shard']['ms_per_step'], 'cam', d['dropin_trainable_camera_step']['leaf_camera_tensors']['ms_per_step'], 'strand', d['strand_stage']['ms_per_iteration_fused'], 'gc', d['host_gc'])" | tee -a $O/gc.log
done; done
