#!/bin/bash
# round 6: is the config5_2M block of the default bench line reproducible (3.0 ms once, 2.0-2.1 in every other run)?
cd ${GRAFT_REPO_ROOT:-/root/repo}
R=$PWD; O=$PWD/gpurun_out/r06c5; mkdir -p $O; export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_strand_build.py tests/test_gpu_loss_adam.py tests/test_gpu_fused.py tests/test_gpu_hair_fullsize.py -m gpu -x -q 2>&1 | tail -5 | tee $O/pytest.log
rm -f $O/c5.log
B="python $R/bench.py --no-cpu-baseline --no-op-only --no-camera-block --no-strand-block"
for rep in 1 2; do
$B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[mid + big] headline', d['ms_per_step'], 'config5_2M', d['config5_2M']['ms_per_step'], 'shard', d['config4_shard']['ms_per_step'])" | tee -a $O/c5.log
GHR_NO_SORT_MID=1 $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[big only] headline', d['ms_per_step'], 'config5_2M', d['config5_2M']['ms_per_step'], 'shard', d['config4_shard']['ms_per_step'])" | tee -a $O/c5.log
done
for rep in 1 2; do
python build/head_tree/tools/strandstep.py 40 2>&1 | grep "ms per" | sed 's/^/[848c0c9] /' | tee -a $O/c5.log
python tools/strandstep.py 40 2>&1 | grep "ms per" | sed 's/^/[tree] /' | tee -a $O/c5.log
done
