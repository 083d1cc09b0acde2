#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fused.py tests/test_gpu_loss_adam.py -x -q -m gpu 2>&1 | tail -2
timeout 300 python bench.py --no-cpu-baseline --no-op-only 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('step', d['ms_per_step'], 'shard4', d['config4_shard']['ms_per_step'], 'K8', d['kernels_ms'])"
