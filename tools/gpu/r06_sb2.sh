#!/bin/bash
# round 6: k_tile_sort_big variants on the workloads that launch it (strand stage, cfg5) + the tests the first run failed / added
cd ${GRAFT_REPO_ROOT:-/root/repo}
R=$PWD; O=$PWD/gpurun_out/r06sb2; mkdir -p $O; export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_gpu_hair_fullsize.py tests/test_gpu_loss_adam.py tests/test_strand_build.py -m gpu -x -q 2>&1 | tail -30 | tee $O/pytest.log
rm -f $O/strand_ab.log $O/cfg5.log
kt() {  # $1 = label, $2 = lib, rest = command
  local lab=$1 lib=$2; shift 2
  ( cd /tmp && rm -rf /tmp/p_$lab && GHR_LIB_PATH=$lib timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_$lab -o kt -- "$@" ) > $O/kt_$lab.log 2>&1
  python - <<PY
import csv, glob
for f in glob.glob('/tmp/p_$lab/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'tile_sort' in r['Name'] or 'k_scatter' in r['Name']:
            print('[$lab] KT %-40s calls %5s avg %9.1f us' % (r['Name'][:40], r['Calls'], float(r['AverageNs']) / 1e3))
PY
}
for v in oldsort sb_r3w4 sb_r2w4 sb_r3w8 sb_r2w8; do
  L=$R/build/variants/libghr_$v.so
  GHR_LIB_PATH=$L python tools/strandstep.py 40 2>&1 | grep "ms per" | sed "s/^/[$v] /" | tee -a $O/strand_ab.log
  kt strand_$v $L python $R/tools/strandstep.py 12 | tee -a $O/strand_ab.log
done
for v in oldsort sb_r3w4; do
  L=$R/build/variants/libghr_$v.so
  kt cfg5_$v $L python $R/bench.py --workload cfg5 --steps 10 --warmup 3 --no-cpu-baseline --no-op-only --no-2m --no-camera-block --no-strand-block | tee -a $O/cfg5.log
done
