#!/bin/bash
O=gpurun_out/r4k; mkdir -p $O; export TMPDIR=/tmp PYTHONUNBUFFERED=1
R=$GRAFT_REPO_ROOT
B="python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-op-only --streams 1 --shard-views 0"
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_kt -o kt -- $B ) > $O/kt.log 2>&1
python - <<PY
import csv, glob
for f in glob.glob('/tmp/prof_kt/**/*kernel_stats.csv', recursive=True):
    rows = sorted(csv.DictReader(open(f)), key=lambda r: -float(r.get('TotalDurationNs', 0) or 0))
    for r in rows[:11]:
        print('KT %-50s calls %5s avg %9.1f us' % (r['Name'][:50], r['Calls'], float(r['AverageNs']) / 1e3))
PY
