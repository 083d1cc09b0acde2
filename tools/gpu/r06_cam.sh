#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
R=$PWD; O=$PWD/gpurun_out/r06cam; mkdir -p $O; export TMPDIR=/tmp PYTHONUNBUFFERED=1
for rep in 1 2; do for mode in plain leaf; do python tools/camstep.py $mode 60 2>&1 | grep CAMSTEP | tee -a $O/camstep.log; done; done
( cd /tmp && rm -rf /tmp/prof_leaf && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_leaf -o kt -- python $R/tools/camstep.py leaf 40 ) > $O/kt_leaf.log 2>&1
python - <<PY | tee $O/kt_leaf.txt
import csv, glob
for f in glob.glob('/tmp/prof_leaf/**/*kernel_stats.csv', recursive=True):
    rows = sorted(csv.DictReader(open(f)), key=lambda r: -float(r.get('TotalDurationNs', 0) or 0))
    for r in rows[:14]:
        print('KT %-70s calls %5s avg %9.1f us' % (r['Name'][:70], r['Calls'], float(r['AverageNs']) / 1e3))
PY
