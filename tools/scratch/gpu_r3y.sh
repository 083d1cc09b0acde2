#!/bin/bash
O=gpurun_out/r3y; mkdir -p $O; export TMPDIR=/tmp PYTHONUNBUFFERED=1
R=$GRAFT_REPO_ROOT
for v in ; do
( GHR_K8=$v timeout 300 python tools/bench_hair.py 30000 100000 ) 2>&1 | grep "HAIR fused" | sed "s/^/$v: /"
done
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/hair_kt -o kt -- python $R/tools/bench_hair.py 30000 100000 ) > $O/kt.log 2>&1
python - <<PY
import csv, glob
for f in glob.glob('/tmp/hair_kt/**/*kernel_stats.csv', recursive=True):
    rows = sorted(csv.DictReader(open(f)), key=lambda r: -float(r.get('TotalDurationNs', 0) or 0))
    for r in [x for x in rows if 'ghr::' in x['Name']][:16]:
        print('KT %-60s calls %5s avg %9.1f us' % (r['Name'][:60], r['Calls'], float(r['AverageNs']) / 1e3))
PY
