#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
R=$PWD; P=$PWD/gpurun_out/r06i; mkdir -p $P; export TMPDIR=/tmp PYTHONUNBUFFERED=1
python tools/camstep.py leaf 60 2>&1 | grep CAMSTEP | sed "s/^/[fused adam] /" | tee -a $P/camstep.log
GHR_FUSE_ADAM=0 python tools/camstep.py leaf 60 2>&1 | grep CAMSTEP | sed "s/^/[separate adam] /" | tee -a $P/camstep.log
( cd /tmp && rm -rf /tmp/prof_f && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_f -o kt -- python $R/tools/camstep.py leaf 40 ) > $P/kt.log 2>&1
python - <<PY
import csv, glob
for f in glob.glob('/tmp/prof_f/**/*kernel_stats.csv', recursive=True):
    rows = sorted(csv.DictReader(open(f)), key=lambda r: -float(r.get('TotalDurationNs', 0) or 0))
    for r in rows[:14]:
        print('KT %-64s calls %5s avg %9.1f us tot %8.2f ms' % (r['Name'][:64], r['Calls'], float(r['AverageNs']) / 1e3, float(r['TotalDurationNs']) / 1e6))
PY
