#!/bin/bash
# round 6: k_tile_sort_mid at eight waves per SIMD, split by list length (256 threads up to 2048 keys, 512 up to 4096) or not
cd ${GRAFT_REPO_ROOT:-/root/repo}
R=$PWD; O=$PWD/gpurun_out/r06sm4; mkdir -p $O; export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_reference_live.py tests/test_gpu_hair_fullsize.py -m gpu -x -q 2>&1 | tail -4 | tee $O/pytest.log
GHR_TILE_ORDER=2 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_hair_fullsize.py -m gpu -x -q -k "long_tile_list or strand_stage_size" 2>&1 | tail -2 | tee -a $O/pytest.log
GHR_LIB_PATH=$R/build/variants/libghr_mid_nosplit.so timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "long_tile_list" 2>&1 | tail -2 | tee -a $O/pytest.log
rm -f $O/ab.log
kt() {  # $1 = label, $2 = lib, rest = command
  local lab=$1 lib=$2; shift 2
  ( cd /tmp && rm -rf /tmp/p_$lab && GHR_LIB_PATH=$lib timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_$lab -o kt -- "$@" ) > $O/kt_$lab.log 2>&1
  python - <<PY
import csv, glob
for f in glob.glob('/tmp/p_$lab/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'tile_sort' in r['Name']:
            print('[$lab] KT %-46s calls %5s avg %9.1f us' % (r['Name'][:46], r['Calls'], float(r['AverageNs']) / 1e3))
PY
}
B="python $R/bench.py --workload cfg5 --steps 10 --warmup 3 --no-cpu-baseline --no-op-only --no-2m --no-camera-block --no-strand-block"
for rep in 1 2; do
for v in split mid_nosplit; do
  L=$R/build/variants/libghr_$v.so; [ $v = split ] && L=$R/gaussianhaircut_amd/csrc/libghr_hip.so
  kt strand_$v $L python $R/tools/strandstep.py 12 | tee -a $O/ab.log
  kt cfg5_$v $L $B | tee -a $O/ab.log
done; done
for v in split mid_nosplit; do
  L=$R/build/variants/libghr_$v.so; [ $v = split ] && L=$R/gaussianhaircut_amd/csrc/libghr_hip.so
  GHR_LIB_PATH=$L python tools/strandstep.py 40 2>&1 | grep "ms per" | sed "s/^/[$v] /" | tee -a $O/ab.log
  GHR_LIB_PATH=$L $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[$v] cfg5 ms_per_step', d['ms_per_step'])" | tee -a $O/ab.log
done
