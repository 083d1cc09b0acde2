#!/bin/bash
# per-kernel averages of the single-view step on one stream under rocprofv3 --kernel-trace: tools/gpu/kt.sh [tag] [env...]
cd ${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-kt}; shift
P=$PWD/gpurun_out/kt; mkdir -p $P; export TMPDIR=/tmp PYTHONUNBUFFERED=1
R=$PWD
B="python $R/bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-op-only --streams 1 --shard-views 0"
( cd /tmp && rm -rf /tmp/prof_$TAG && env "$@" timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -o kt -- $B ) > $P/${TAG}.log 2>&1; echo "kt rc=$?"
python - <<PY
import csv, glob
for f in glob.glob('/tmp/prof_$TAG/**/*kernel_stats.csv', recursive=True):
    rows = sorted(csv.DictReader(open(f)), key=lambda r: -float(r.get('TotalDurationNs', 0) or 0))
    for r in rows[:14]:
        print('KT[$TAG] %-52s calls %5s avg %9.1f us' % (r['Name'][:52], r['Calls'], float(r['AverageNs']) / 1e3))
PY
grep -o '"ms_per_step": [0-9.]*' $P/${TAG}.log | head -2
