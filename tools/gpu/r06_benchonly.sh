#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
P=$PWD/gpurun_out/profiles; mkdir -p $P; export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > /dev/null 2>&1
timeout 600 python bench.py --steps 20 --warmup 5 > $P/r06_bench.json 2> $P/r06_bench.err; echo "bench rc=$?"
python - <<PY
import json
b=json.load(open('$P/r06_bench.json'))
print({k:b[k] for k in ('value','ms_per_step')}, b['roofline']['frac'], b['roofline']['avg_kernel_ms'])
for k in ('fixed_camera_step','densify_stats_step'): print(k, b[k])
print(b['config4_shard']['ms_per_step'], b['config5_2M']['ms_per_step'], b['config5_2M']['k_render_bwd_ms'], b['dropin_trainable_camera_step']['leaf_camera_tensors'], b['dropin_trainable_camera_step']['residual_parameters']['ms_per_step'], b['strand_stage']['ms_per_iteration_fused'])
print({k:(v['fwd_ms'],v['bwd_ms']) for k,v in b['op_only'].items()}, b['cpu_baseline']['value'])
PY
