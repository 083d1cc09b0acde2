#!/bin/bash
# per-kernel averages of the rasterizer op alone (bench.op_only_bench) under rocprofv3: tools/gpu/opstats.sh cfg2|cfg3
cd ${GRAFT_REPO_ROOT:-/root/repo}; export TMPDIR=/tmp
cfg=${1:-cfg2}
R=$PWD
cd /tmp && rm -rf /tmp/opk && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/opk -o kt -- python -c "
import sys; sys.path.insert(0, '$R')
import torch, bench
print(bench.op_only_bench(torch.device('cuda:0'), '$cfg', iters=30, warm=5))" 2>&1 | grep "fwd_ms" | cut -c1-300
python - <<PY
import csv, glob
for f in glob.glob("/tmp/opk/**/*kernel_stats.csv", recursive=True):
    rows = sorted(csv.DictReader(open(f)), key=lambda r: -float(r.get("TotalDurationNs", 0) or 0))
    for r in rows[:14]:
        print("KT[$cfg] %-56s calls %5s avg %9.1f us" % (r["Name"][:56], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
