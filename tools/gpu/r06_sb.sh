#!/bin/bash
# round 6, strand stage: k_tile_sort_big on the register-blocked network, compact per-row outputs, out-of-place late groups
cd ${GRAFT_REPO_ROOT:-/root/repo}
R=$PWD; O=$PWD/gpurun_out/r06sb; mkdir -p $O; export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_reference_live.py tests/test_gpu_fused.py tests/test_reference_dropin.py tests/test_camera_grads.py tests/test_gpu_hair_fullsize.py tests/test_strand_build.py tests/test_gpu_loss_adam.py -m gpu -x -q 2>&1 | tail -8 | tee $O/pytest.log
rm -f $O/strand_ab.log
for rep in 1 2; do
GHR_LIB_PATH=$R/build/variants/libghr_oldsort.so python tools/strandstep.py 40 2>&1 | grep STRAND | sed 's/^/[old sort_big] /' | tee -a $O/strand_ab.log
python tools/strandstep.py 40 2>&1 | grep STRAND | sed 's/^/[blocked sort_big] /' | tee -a $O/strand_ab.log
done
for v in oldsort new; do
  L=$R/gaussianhaircut_amd/csrc/libghr_hip.so; [ $v = oldsort ] && L=$R/build/variants/libghr_oldsort.so
  ( cd /tmp && rm -rf /tmp/sc_$v && GHR_LIB_PATH=$L timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/sc_$v -- python $R/tools/sort_cliff.py ) > $O/sc_$v.log 2>&1
  echo "== sort_cliff $v" | tee -a $O/sort_cliff.txt
  python tools/sort_cliff.py --parse /tmp/sc_$v | tee -a $O/sort_cliff.txt
done
( cd /tmp && rm -rf /tmp/prof_s && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_s -o kt -- python $R/tools/strandstep.py 20 ) > $O/kt.log 2>&1
python - <<PY | tee $O/kt_summary.txt
import csv, glob
for f in glob.glob('/tmp/prof_s/**/*kernel_stats.csv', recursive=True):
    rows = sorted(csv.DictReader(open(f)), key=lambda r: -float(r.get('TotalDurationNs', 0) or 0))
    for r in rows[:30]:
        print('KT %-70s calls %5s avg %9.1f us' % (r['Name'][:70], r['Calls'], float(r['AverageNs']) / 1e3))
PY
