#!/bin/bash
# PROBE: the two loss kernels with the traffic of the nine scratch maps removed (results wrong): the HBM-side ceiling of a single-pass loss
cd ${GRAFT_REPO_ROOT:-/root/repo}
R=$PWD; P=$PWD/gpurun_out/r06u; mkdir -p $P; export TMPDIR=/tmp PYTHONUNBUFFERED=1
for v in product nomaps product nomaps; do
  lib=""; [ "$v" != product ] && lib=$R/build/variants/libghr_$v.so
  ( cd /tmp && rm -rf /tmp/prof_l && GHR_LIB_PATH=$lib timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_l -o kt -- python $R/tools/camstep.py fixed 40 ) > $P/kt_$v.log 2>&1
  python - <<PY | tee -a $P/loss_probe.txt
import csv, glob
for f in glob.glob('/tmp/prof_l/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'k_loss_fwd' in r['Name'] or 'k_loss_bwd' in r['Name']:
            print('LOSS [$v] %-44s calls %4s avg %7.1f us' % (r['Name'][:44], r['Calls'], float(r['AverageNs']) / 1e3))
PY
done
