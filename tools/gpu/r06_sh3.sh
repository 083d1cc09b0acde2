#!/bin/bash
# round 6: the N > 1 bench path with the factored SH message: 2 ranks sharing the GPU over gloo (functional), 1 rank through RCCL
cd ${GRAFT_REPO_ROOT:-/root/repo}
P=$PWD/gpurun_out/r06sh; mkdir -p $P; export TMPDIR=/tmp PYTHONUNBUFFERED=1
GHR_BENCH_BACKEND=gloo GHR_BENCH_SHARE_GPU=1 timeout 900 python bench.py --gpus 2 --steps 6 --warmup 2 > $P/bench_2ranks_gloo_shared.json 2> $P/bench_2ranks.err; echo "2-rank rc=$?"
python - <<PY
import json
d = json.load(open("$P/bench_2ranks_gloo_shared.json"))
print("n_gpus", d["n_gpus"], "replicas_identical", d.get("replicas_identical"), "ms", d["ms_per_step"])
print(json.dumps(d.get("scaling_breakdown"))[:2500])
PY
tail -c 400 $P/bench_2ranks.err
GHR_FORCE_COLLECTIVES=1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29571 bench.py --gpus 1 --steps 10 --warmup 3 --no-2m --no-camera-block --no-strand-block --no-op-only --no-cpu-baseline > $P/bench_1rank_rccl_forced.json 2> $P/bench_1rank.err; echo "1-rank rccl rc=$?"
python - <<PY
import json
d = json.load(open("$P/bench_1rank_rccl_forced.json"))
print("n_gpus", d["n_gpus"], "ms", d["ms_per_step"], d["config"].get("optimizer"))
print(json.dumps(d.get("scaling_breakdown"))[:2500])
PY
tail -c 400 $P/bench_1rank.err
