#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
R=$PWD; P=$PWD/gpurun_out/r06m; mkdir -p $P; export TMPDIR=/tmp PYTHONUNBUFFERED=1
cat > /tmp/oponly.py <<PY
import sys, json, torch
sys.path.insert(0, "$R")
import bench
print("OPONLY", json.dumps({k: v for k, v in bench.op_only_bench(torch.device("cuda:0"), "cfg3", iters=30).items() if k in ("fwd_ms", "bwd_ms", "bwd_ms_host_sync_every_iter")}))
PY
python /tmp/oponly.py 2>&1 | grep OPONLY | tee $P/oponly.log
( cd /tmp && rm -rf /tmp/prof_o && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_o -o kt -- python /tmp/oponly.py ) > $P/kt.log 2>&1
python - <<PY
import csv, glob
for f in glob.glob('/tmp/prof_o/**/*kernel_stats.csv', recursive=True):
    rows = sorted(csv.DictReader(open(f)), key=lambda r: -float(r.get('TotalDurationNs', 0) or 0))
    for r in rows[:14]:
        print('KT %-64s calls %5s avg %9.1f us tot %8.2f ms' % (r['Name'][:64], r['Calls'], float(r['AverageNs']) / 1e3, float(r['TotalDurationNs']) / 1e6))
PY
