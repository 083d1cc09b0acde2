#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
P=$PWD/gpurun_out/r06o; mkdir -p $P; export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
T0=$(date +%s); timeout 900 python bench.py > $P/bench.json 2> $P/bench.err; echo "bench rc=$? wall $(( $(date +%s) - T0 )) s"; tail -3 $P/bench.err
python - <<PY
import json
d = json.load(open("$P/bench.json"))
print("ms_per_step", d["ms_per_step"], "frac", d["roofline"]["frac"])
for k in ("fixed_camera_step", "densify_stats_step", "config4_shard", "config5_2M", "dropin_trainable_camera_step", "strand_stage"):
    print(k, json.dumps(d.get(k))[:500])
print({c: (d["op_only"][c]["fwd_ms"], d["op_only"][c]["bwd_ms"]) for c in d["op_only"]})
PY
