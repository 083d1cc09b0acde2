#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
R=$PWD; O=$PWD/gpurun_out/r06hf; mkdir -p $O; export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_gpu_fused.py tests/test_reference_dropin.py tests/test_camera_grads.py tests/test_gpu_hair_fullsize.py tests/test_strand_build.py -m gpu -x -q -k "hair or strand" 2>&1 | tail -3 | tee $O/pytest_hair.log
rm -f $O/strand_ab.log
for rep in 1 2; do
GHR_FUSE_STRAND_ADAM=0 python tools/strandstep.py 40 2>&1 | grep STRAND | sed 's/^/[separate Adam pass] /' | tee -a $O/strand_ab.log
GHR_FUSE_STRAND_ADAM=1 python tools/strandstep.py 40 2>&1 | grep STRAND | sed 's/^/[update in the backward] /' | tee -a $O/strand_ab.log
done
( cd /tmp && rm -rf /tmp/prof_s && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_s -o kt -- python $R/tools/strandstep.py 20 ) > $O/kt.log 2>&1
python - <<PY | tee $O/kt_summary.txt
import csv, glob
for f in glob.glob('/tmp/prof_s/**/*kernel_stats.csv', recursive=True):
    rows = sorted(csv.DictReader(open(f)), key=lambda r: -float(r.get('TotalDurationNs', 0) or 0))
    for r in rows[:16]:
        print('KT %-70s calls %5s avg %9.1f us' % (r['Name'][:70], r['Calls'], float(r['AverageNs']) / 1e3))
PY
