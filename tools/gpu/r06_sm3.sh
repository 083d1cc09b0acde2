#!/bin/bash
# round 6: k_tile_sort_mid at four workgroups per CU (64 VGPRs, one spilled) / with 1024 workgroups, against three per CU and 768
cd ${GRAFT_REPO_ROOT:-/root/repo}
R=$PWD; O=$PWD/gpurun_out/r06sm3; mkdir -p $O; export TMPDIR=/tmp PYTHONUNBUFFERED=1
rm -f $O/ab.log
kt() {  # $1 = label, $2 = lib, rest = command
  local lab=$1 lib=$2; shift 2
  ( cd /tmp && rm -rf /tmp/p_$lab && GHR_LIB_PATH=$lib timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_$lab -o kt -- "$@" ) > $O/kt_$lab.log 2>&1
  python - <<PY
import csv, glob
for f in glob.glob('/tmp/p_$lab/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'tile_sort_mid' in r['Name']:
            print('[$lab] KT %-40s calls %5s avg %9.1f us' % (r['Name'][:40], r['Calls'], float(r['AverageNs']) / 1e3))
PY
}
B="python $R/bench.py --workload cfg5 --steps 10 --warmup 3 --no-cpu-baseline --no-op-only --no-2m --no-camera-block --no-strand-block"
for rep in 1 2; do
for v in default mid_w8_g1024 mid_w8_g768 mid_w6_g1024; do
  L=$R/build/variants/libghr_$v.so; [ $v = default ] && L=$R/gaussianhaircut_amd/csrc/libghr_hip.so
  kt strand_$v $L python $R/tools/strandstep.py 12 | tee -a $O/ab.log
  kt cfg5_$v $L $B | tee -a $O/ab.log
done; done
