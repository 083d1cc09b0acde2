#!/bin/bash
# per-kernel averages of the bench step (GPU box tool); every command is time-boxed
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/kts; ( cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kts -o kt -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-op-only ${BENCH_ARGS---streams 1} > /dev/null 2>&1 )
python - <<PY
import csv,glob
for f in glob.glob('/tmp/kts/**/*kernel_stats.csv', recursive=True):
    rows=[r for r in csv.DictReader(open(f)) if 'ghr::' in r['Name'] or 'k_loss' in r['Name']]
    print('KSTATS', ' '.join('%s=%.1f' % (r['Name'].split('(')[0].replace('ghr::k_',''), float(r['AverageNs'])/1e3) for r in rows))
PY
