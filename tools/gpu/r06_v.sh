#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
R=$PWD; P=$PWD/gpurun_out/r06v; mkdir -p $P; export TMPDIR=/tmp PYTHONUNBUFFERED=1
python tools/strandstep.py 20 2>&1 | grep STRAND | tee $P/strand.log
( cd /tmp && rm -rf /tmp/prof_s && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_s -o kt -- python $R/tools/strandstep.py 20 ) > $P/kt.log 2>&1
python - <<PY | tee $P/kt_summary.txt
import csv, glob
for f in glob.glob('/tmp/prof_s/**/*kernel_stats.csv', recursive=True):
    rows = sorted(csv.DictReader(open(f)), key=lambda r: -float(r.get('TotalDurationNs', 0) or 0))
    tot = sum(float(r['TotalDurationNs']) for r in rows)
    print('KT total %.1f ms over %d launches' % (tot / 1e6, sum(int(r['Calls']) for r in rows)))
    for r in rows[:40]:
        print('KT %-84s calls %5s avg %9.1f us tot %8.2f ms' % (r['Name'][:84], r['Calls'], float(r['AverageNs']) / 1e3, float(r['TotalDurationNs']) / 1e6))
PY
