#!/bin/bash
# round 6, trip 5: reworked camera-gradient kernels (tests + step cost), norm probe, densification statistics, classic K8 forms on cfg2
cd ${GRAFT_REPO_ROOT:-/root/repo}
R=$PWD; P=$PWD/gpurun_out/r06e; mkdir -p $P; export TMPDIR=/tmp PYTHONUNBUFFERED=1
python tools/gpu/norm_probe.py 2>&1 | grep NORM | tee $P/norm_probe.txt
timeout 900 python -m pytest tests/test_gpu_fused.py tests/test_camera_grads.py -m gpu -q 2>&1 | tail -12 | tee $P/pytest.log
for mode in plain leaf; do python tools/camstep.py $mode 60 2>&1 | grep CAMSTEP | tee -a $P/camstep.log; done
for mode in leaf; do
  ( cd /tmp && rm -rf /tmp/prof_$mode && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$mode -o kt -- python $R/tools/camstep.py $mode 40 ) > $P/kt_$mode.log 2>&1
  python - <<PY
import csv, glob
for f in glob.glob('/tmp/prof_$mode/**/*kernel_stats.csv', recursive=True):
    rows = sorted(csv.DictReader(open(f)), key=lambda r: -float(r.get('TotalDurationNs', 0) or 0))
    for r in rows[:30]:
        print('KT[$mode] %-64s calls %5s avg %9.1f us tot %8.2f ms' % (r['Name'][:64], r['Calls'], float(r['AverageNs']) / 1e3, float(r['TotalDurationNs']) / 1e6))
PY
done 2>&1 | tee $P/kt_summary.log | grep "project\|cam_fold"
for cfg in cfg2 cfg3; do
  timeout 120 python tools/kbench.py $cfg 30 2>&1 | grep KBENCH | sed "s/^/[cells] /" | tee -a $P/kbench_k8_forms.log
  GHR_K8=cell timeout 120 python tools/kbench.py $cfg 30 2>&1 | grep KBENCH | sed "s/^/[GHR_K8=cell] /" | tee -a $P/kbench_k8_forms.log
done
timeout 600 python tools/densify_bench.py cfg3 2>&1 | grep DENSIFY | tee $P/densify.txt
