#!/bin/bash
# round 6: HBM traffic of the strand stage's projection backward (k_project_bwd<false, true>, mode 1 + SH update) against its byte count
cd ${GRAFT_REPO_ROOT:-/root/repo}
R=$PWD; O=$PWD/gpurun_out/r06pbs; mkdir -p $O; export TMPDIR=/tmp PYTHONUNBUFFERED=1
for ctr in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && rm -rf /tmp/pmc_s_$ctr && timeout 300 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d /tmp/pmc_s_$ctr -o p -- python $R/tools/strandstep.py 6 ) > $O/pmc_$ctr.log 2>&1; echo "pmc $ctr rc=$?"
done
python - <<'PY' | tee $O/traffic.txt
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for ctr in ('FETCH_SIZE', 'WRITE_SIZE'):
    for f in glob.glob('/tmp/pmc_s_%s/**/*counter_collection.csv' % ctr, recursive=True):
        for r in csv.DictReader(open(f)):
            if r['Counter_Name'] == ctr:
                acc[r['Kernel_Name'].split('(')[0][-48:]][ctr].append(float(r['Counter_Value']))
for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1].get('FETCH_SIZE', [0]))):
    f = v.get('FETCH_SIZE', []); w = v.get('WRITE_SIZE', [])
    if not f or not w: continue
    fm, wm = sum(f[-4:]) / len(f[-4:]), sum(w[-4:]) / len(w[-4:])
    if (2 * fm + wm) * 1024 < 20e6: continue
    print('%-50s launches %3d  FETCH %8.1f MB (x2 = %8.1f)  WRITE %8.1f MB  -> %8.1f MB' % (k, len(f), fm * 1024 / 1e6, 2 * fm * 1024 / 1e6, wm * 1024 / 1e6, (2 * fm + wm) * 1024 / 1e6))
PY
