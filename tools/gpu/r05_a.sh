#!/bin/bash
# round 5, first GPU pass: the whole -m gpu suite, the bench line, per-kernel averages of the step
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=$PWD/gpurun_out/r05a; mkdir -p $O; export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -40 > $O/pytest.log; tail -15 $O/pytest.log
timeout 400 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python - <<PY
import json
d = json.loads(open('$O/bench.json').read().strip().splitlines()[-1])
print('STEP ms', d['ms_per_step'], 'K8', d['kernels_ms'], 'roof', d['roofline']['frac'], 'cfg4', d['config4_shard']['ms_per_step'])
for k, v in d['op_only'].items(): print('OP', k, v['fwd_ms'], v['bwd_ms'], v['whole_forward_hbm_frac'], v['whole_backward_hbm_frac'])
PY
bash tools/gpu/kt.sh r05a
