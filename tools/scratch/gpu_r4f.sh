#!/bin/bash
O=gpurun_out/r4f; mkdir -p $O; export TMPDIR=/tmp PYTHONUNBUFFERED=1
R=$GRAFT_REPO_ROOT
for rep in 1 2 3; do
timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-op-only 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('STEP ms_per_step %.4f shard4 %.4f K7 %.4f K8 %.4f frac %.4f' % (d['ms_per_step'], d['config4_shard']['ms_per_step'], d['kernels_ms']['k_render_fwd'], d['kernels_ms']['k_render_bwd'], d['roofline']['frac']))"
done
B="python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-op-only --streams 1 --shard-views 0"
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_kt -o kt -- $B ) > $O/kt.log 2>&1
python - <<PY
import csv, glob
for f in glob.glob('/tmp/prof_kt/**/*kernel_stats.csv', recursive=True):
    rows = sorted(csv.DictReader(open(f)), key=lambda r: -float(r.get('TotalDurationNs', 0) or 0))
    for r in rows[:11]:
        print('KT %-50s calls %5s avg %9.1f us' % (r['Name'][:50], r['Calls'], float(r['AverageNs']) / 1e3))
PY
