#!/bin/bash
# round 5: the tile sort as the prologue of K7's workgroups (default) against the kernel of its own (GHR_FUSE_SORT=0), same box
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=$PWD/gpurun_out/r05h; mkdir -p $O; export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 400 python -m pytest tests/test_gpu_parity.py tests/test_reference_cuda_golden.py tests/test_gpu_reference_live.py -m gpu -q -x 2>&1 | tail -4
bash tools/gpu/kt.sh h_fused 2>&1 | grep -E "k_render_fwd|k_tile_sort|k_render_bwd|k_loss_fwd|ms_per_step"
bash tools/gpu/kt.sh h_split GHR_FUSE_SORT=0 2>&1 | grep -E "k_render_fwd|k_tile_sort|k_render_bwd|k_loss_fwd|ms_per_step"
timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench.json 2>/dev/null; python - <<PY
import json
d = json.loads(open('$O/bench.json').read().strip().splitlines()[-1])
print('FUSED STEP ms', d['ms_per_step'], 'cfg4', d['config4_shard']['ms_per_step'], 'K8', d['kernels_ms'])
for k, v in d['op_only'].items(): print('   OP', k, v['fwd_ms'], v['bwd_ms'], v['whole_forward_hbm_frac'], v['whole_backward_hbm_frac'])
PY
