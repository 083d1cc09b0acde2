#!/bin/bash
# round 6, trip 3: (a) K8 against its LDS footprint: 5 / 4 / 3 / 2 workgroups per CU (kbench, cfg2 + cfg3); (b) kernel traces of
# the step: plain, fixed camera, leaf camera, residual camera
cd ${GRAFT_REPO_ROOT:-/root/repo}
R=$PWD; P=$PWD/gpurun_out/r06c; mkdir -p $P; export TMPDIR=/tmp PYTHONUNBUFFERED=1
for cfg in cfg3 cfg2; do
for v in product pad4wg pad3wg pad2wg product; do
  lib=""; [ "$v" != product ] && lib=$PWD/build/variants/libghr_$v.so
  GHR_LIB_PATH=$lib timeout 120 python tools/kbench.py $cfg 30 2>&1 | grep "KBENCH" | sed "s/^/[$v] /" | tee -a $P/kbench_lds.log
done; done
for mode in plain fixed leaf residual; do
  python tools/camstep.py $mode 40 2>&1 | grep CAMSTEP | tee -a $P/camstep.log
  ( cd /tmp && rm -rf /tmp/prof_$mode && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$mode -o kt -- python $R/tools/camstep.py $mode 40 ) > $P/kt_$mode.log 2>&1
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
    for r in rows[:22]:
        print('KT[$mode] %-64s calls %5s avg %9.1f us tot %8.2f ms' % (r['Name'][:64], r['Calls'], float(r['AverageNs']) / 1e3, float(r['TotalDurationNs']) / 1e6))
PY
done 2>&1 | tee $P/kt_summary.log | grep -v "^KT\[.*\] .*avg" | tail -20
