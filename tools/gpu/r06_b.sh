#!/bin/bash
# round 6, second trip: fixed GPU tests, the new bench line (cycling cameras, 2M block, trainable-camera block), self-launched 2 ranks
cd ${GRAFT_REPO_ROOT:-/root/repo}
P=$PWD/gpurun_out/r06b; mkdir -p $P; export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_gpu_fused.py tests/test_camera_grads.py -m gpu -q 2>&1 | tail -30 > $P/pytest.log; tail -5 $P/pytest.log
timeout 900 python bench.py > $P/bench.json 2> $P/bench.err; echo "bench rc=$?"; tail -c 300 $P/bench.err
python - <<PY
import json
d = json.load(open("$P/bench.json"))
print("ms_per_step", d["ms_per_step"], "value", d["value"], "frac", d["roofline"]["frac"], d["roofline"]["avg_kernel_ms"], d["roofline"]["camera0"])
for k in ("fixed_camera_step", "config4_shard", "config5_2M", "dropin_trainable_camera_step"):
    print(k, json.dumps(d.get(k))[:900])
print("P_vis", d["config"]["P_visible_per_camera"]); print("R", d["config"]["num_rendered_per_camera"])
PY
GHR_BENCH_BACKEND=gloo GHR_BENCH_SHARE_GPU=1 timeout 900 python bench.py --gpus 2 --steps 6 --warmup 2 > $P/bench_2ranks_gloo_shared.json 2> $P/bench_2ranks.err; echo "2-rank rc=$?"
python - <<PY
import json
d = json.load(open("$P/bench_2ranks_gloo_shared.json"))
print("n_gpus", d["n_gpus"], "replicas_identical", d.get("replicas_identical"), "ms", d["ms_per_step"], d["config"]["workload"][-120:])
print(json.dumps(d.get("scaling_breakdown"))[:1200])
PY
tail -c 600 $P/bench_2ranks.err
