#!/bin/bash
# round 6: config5_2M block of the default bench line with the capacity guess on a grid (allocation sizes repeat), four runs
cd ${GRAFT_REPO_ROOT:-/root/repo}
R=$PWD; O=$PWD/gpurun_out/r06c5b; mkdir -p $O; export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 600 python -m pytest tests/test_gpu_fused.py tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -3 | tee $O/pytest.log
rm -f $O/c5.log
B="python $R/bench.py --no-cpu-baseline --no-op-only --no-camera-block"
for rep in 1 2 3 4; do
$B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[capacity on a grid] headline', d['ms_per_step'], 'fixed', d['fixed_camera_step']['ms_per_step'], 'config5_2M', d['config5_2M']['ms_per_step'], 'shard', d['config4_shard']['ms_per_step'], 'strand', d['strand_stage']['ms_per_iteration_fused'])" | tee -a $O/c5.log
done
