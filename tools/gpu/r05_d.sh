#!/bin/bash
# round 5: full -m gpu suite on HEAD (loss slots, recycled image workspace, binning changes), then per-tile counters
# padded to one 64-B / 128-B line each (variants tc16 / tc32) against the packed ones, same box
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=$PWD/gpurun_out/r05d; mkdir -p $O; export TMPDIR=/tmp PYTHONUNBUFFERED=1
V=$PWD/build/variants
timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -12 > $O/pytest.log; tail -6 $O/pytest.log
for rep in 1 2; do
  bash tools/gpu/kt.sh head$rep 2>&1 | grep -E "k_project\(|k_scatter|k_tile_scan|k_tile_sort|fillBuffer|k_preprocess|ms_per_step"
  for v in tc16 tc32; do bash tools/gpu/kt.sh $v$rep GHR_LIB_PATH=$V/libghr_$v.so 2>&1 | grep -E "k_project\(|k_scatter|k_tile_scan|k_tile_sort|ms_per_step"; done
done
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python - <<PY
import json
d = json.loads(open('$O/bench.json').read().strip().splitlines()[-1])
print('STEP ms', d['ms_per_step'], 'K8', d['kernels_ms'], 'roof', d['roofline']['frac'], 'stale', d['roofline'].get('traffic_stale'), 'cfg4', d['config4_shard']['ms_per_step'])
for k, v in d['op_only'].items(): print('OP', k, v['fwd_ms'], v['bwd_ms'], v['whole_forward_hbm_frac'], v['whole_backward_hbm_frac'])
PY
