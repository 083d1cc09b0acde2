#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
P=$PWD/gpurun_out/r06p; mkdir -p $P; export TMPDIR=/tmp PYTHONUNBUFFERED=1
GHR_BENCH_BACKEND=gloo GHR_BENCH_SHARE_GPU=1 timeout 900 python bench.py --gpus 2 --steps 6 --warmup 2 > $P/bench_2ranks.json 2> $P/bench_2ranks.err; echo "2-rank rc=$?"
python - <<PY
import json
d = json.loads([l for l in open("$P/bench_2ranks.json") if l.startswith("{")][-1])
print("n_gpus", d["n_gpus"], "replicas_identical", d.get("replicas_identical"), "ms", d["ms_per_step"], d["config"]["optimizer"], list(d.keys()))
PY
GHR_FORCE_COLLECTIVES=1 timeout 600 python bench.py --steps 10 --warmup 3 --no-op-only --no-cpu-baseline --no-2m --no-camera-block --no-strand-block > $P/bench_1rank_forced_collectives.json 2> $P/b.err; echo "forced rc=$?"; python -c "
import json; d=json.load(open('$P/bench_1rank_forced_collectives.json')); print(d['ms_per_step'], d['config']['optimizer'], json.dumps(d.get('scaling_breakdown'))[:400])"
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -6 | tee $P/pytest_all.log
