#!/bin/bash
# round 6, trip 4: LDS accumulation ubench; densification-statistics test; kernel traces of the step (plain / fixed / leaf / residual)
cd ${GRAFT_REPO_ROOT:-/root/repo}
R=$PWD; P=$PWD/gpurun_out/r06d; mkdir -p $P; export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 300 tools/ubench/lds_accum_rate 2>&1 | grep UBENCH > $P/lds_accum_rate.txt; tail -42 $P/lds_accum_rate.txt
timeout 600 python -m pytest tests/test_gpu_fused.py -m gpu -q -k "densification or camera or empty or recycled" 2>&1 | tail -15 | tee $P/pytest.log
for mode in plain fixed leaf; do
  ( cd /tmp && rm -rf /tmp/prof_$mode && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$mode -o kt -- python $R/tools/camstep.py $mode 40 ) > $P/kt_$mode.log 2>&1
  grep CAMSTEP $P/kt_$mode.log
  python - <<PY
import csv, glob
for f in glob.glob('/tmp/prof_$mode/**/*kernel_stats.csv', recursive=True):
    rows = sorted(csv.DictReader(open(f)), key=lambda r: -float(r.get('TotalDurationNs', 0) or 0))
    tot = sum(float(r['TotalDurationNs']) for r in rows)
    print('KT[$mode] total GPU kernel time %.1f ms, %d kernel names, %d launches' % (tot / 1e6, len(rows), sum(int(r['Calls']) for r in rows)))
    with open('$P/kt_${mode}_stats.csv', 'w', newline='') as fo:
        w = csv.DictWriter(fo, fieldnames=rows[0].keys()); w.writeheader()
        for r in rows[:60]:
            r = dict(r); r['Name'] = r['Name'][:100]; w.writerow(r)
    for r in rows[:26]:
        print('KT[$mode] %-64s calls %5s avg %9.1f us tot %8.2f ms' % (r['Name'][:64], r['Calls'], float(r['AverageNs']) / 1e3, float(r['TotalDurationNs']) / 1e6))
PY
done 2>&1 | tee $P/kt_summary.log | grep "total GPU\|CAMSTEP"
