#!/bin/bash
# round 6: k_tile_sort_mid (lists of 1025 .. 4096 keys) -- parity, then the workloads that launch the dense-tile kernels, with / without
cd ${GRAFT_REPO_ROOT:-/root/repo}
R=$PWD; O=$PWD/gpurun_out/r06sm; mkdir -p $O; export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_reference_live.py tests/test_gpu_hair_fullsize.py tests/test_gpu_fused_fullsize.py -m gpu -x -q 2>&1 | tail -12 | tee $O/pytest.log
rm -f $O/ab.log
kt() {  # $1 = label, rest = command
  local lab=$1; shift
  ( cd /tmp && rm -rf /tmp/p_$lab && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_$lab -o kt -- "$@" ) > $O/kt_$lab.log 2>&1
  python - <<PY
import csv, glob
for f in glob.glob('/tmp/p_$lab/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'tile_sort' in r['Name']:
            print('[$lab] KT %-40s calls %5s avg %9.1f us' % (r['Name'][:40], r['Calls'], float(r['AverageNs']) / 1e3))
PY
}
for rep in 1 2; do
GHR_NO_SORT_MID=1 python tools/strandstep.py 40 2>&1 | grep "ms per" | sed "s/^/[big only] /" | tee -a $O/ab.log
python tools/strandstep.py 40 2>&1 | grep "ms per" | sed "s/^/[mid + big] /" | tee -a $O/ab.log
done
GHR_NO_SORT_MID=1 kt strand_bigonly python $R/tools/strandstep.py 12 | tee -a $O/ab.log
kt strand_mid python $R/tools/strandstep.py 12 | tee -a $O/ab.log
B="python $R/bench.py --workload cfg5 --steps 10 --warmup 3 --no-cpu-baseline --no-op-only --no-2m --no-camera-block --no-strand-block"
for rep in 1 2; do
GHR_NO_SORT_MID=1 $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[big only] cfg5 ms_per_step', d['ms_per_step'])" | tee -a $O/ab.log
$B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[mid + big] cfg5 ms_per_step', d['ms_per_step'])" | tee -a $O/ab.log
done
GHR_NO_SORT_MID=1 kt cfg5_bigonly $B | tee -a $O/ab.log
kt cfg5_mid $B | tee -a $O/ab.log
