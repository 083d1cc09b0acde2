#!/bin/bash
# K7 / K8 as they run INSIDE the training step (cold caches behind the loss kernels, sustained clocks) for a list of
# library variants: tools/stepbench.sh [lib.so ...]   ("" = the product build); GHR_K8 is honoured
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
for lib in "$@"; do
  GHR_LIB_PATH=$lib timeout 300 python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-op-only --shard-views 0 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('STEP lib=%s K8=%s  ms_per_step %.4f  k_render_fwd %.4f  k_render_bwd %.4f' % ('$lib'.split('/')[-1] or 'product', '${GHR_K8:-cells}', d['ms_per_step'], d['kernels_ms']['k_render_fwd'], d['kernels_ms']['k_render_bwd']))"
done
