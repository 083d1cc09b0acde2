#!/bin/bash
# round 6, trip 6: Adam fused into the last projection backward: tests, step time, kernel trace; host-side cost of a step
cd ${GRAFT_REPO_ROOT:-/root/repo}
R=$PWD; P=$PWD/gpurun_out/r06f; mkdir -p $P; export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_gpu_fused.py -m gpu -q -x 2>&1 | tail -25 | tee $P/pytest_fused.log
GHR_FUSE_ADAM=0 python tools/camstep.py plain 60 2>&1 | grep CAMSTEP | sed "s/^/[separate adam] /" | tee -a $P/camstep.log
python tools/camstep.py plain 60 2>&1 | grep CAMSTEP | sed "s/^/[fused adam] /" | tee -a $P/camstep.log
GHR_FUSE_ADAM=0 python tools/camstep.py plain 60 2>&1 | grep CAMSTEP | sed "s/^/[separate adam] /" | tee -a $P/camstep.log
python tools/camstep.py plain 60 2>&1 | grep CAMSTEP | sed "s/^/[fused adam] /" | tee -a $P/camstep.log
CAMSTEP_CFG=cfg1 python tools/camstep.py plain 300 2>&1 | grep CAMSTEP | sed "s/^/[cfg1: host-bound] /" | tee -a $P/camstep.log
CAMSTEP_CFG=cfg1 GHR_FUSE_ADAM=0 python tools/camstep.py plain 300 2>&1 | grep CAMSTEP | sed "s/^/[cfg1: host-bound, separate adam] /" | tee -a $P/camstep.log
( cd /tmp && rm -rf /tmp/prof_f && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_f -o kt -- python $R/tools/camstep.py plain 40 ) > $P/kt.log 2>&1
python - <<PY
import csv, glob
for f in glob.glob('/tmp/prof_f/**/*kernel_stats.csv', recursive=True):
    rows = sorted(csv.DictReader(open(f)), key=lambda r: -float(r.get('TotalDurationNs', 0) or 0))
    for r in rows[:16]:
        print('KT %-64s calls %5s avg %9.1f us tot %8.2f ms' % (r['Name'][:64], r['Calls'], float(r['AverageNs']) / 1e3, float(r['TotalDurationNs']) / 1e6))
PY
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -8 | tee $P/pytest_all.log
