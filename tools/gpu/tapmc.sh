#!/bin/bash
# (at most four TCP counters per pass: a fifth makes rocprofv3 abort and hang until its timeout)
# texture-address / L1 (TA, TCP) counters of the render kernels over tools/kbench.py: tools/gpu/tapmc.sh <outfile> [cfg] [ENV=val ...]
cd ${GRAFT_REPO_ROOT:-/root/repo}
out=$1; cfg=${2:-cfg3}; shift; shift
export TMPDIR=/tmp
R=$PWD
i=0
for set in "GRBM_GUI_ACTIVE TA_BUSY_avr TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum SQ_INSTS_VMEM"; do
  i=$((i+1)); rm -rf /tmp/ta$i
  ( cd /tmp && env "$@" timeout 90 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/ta$i -o p -- python $R/tools/kbench.py $cfg 5 ) > /tmp/ta$i.log 2>&1
  python - <<PY >> $out
import csv,glob,collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for fn in glob.glob('/tmp/ta$i/**/*counter_collection.csv',recursive=True):
    for r in csv.DictReader(open(fn)):
        k=r['Kernel_Name'].split('(')[0].replace('ghr::','')
        if k.startswith('k_render'): acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k in sorted(acc):
    print('TAPMC [$cfg $*]', k, ' '.join('%s=%.4g' % (c, sum(v)/len(v)) for c,v in acc[k].items()))
if not acc: print('TAPMC set $i: nothing', open('/tmp/ta$i.log').read()[-400:])
PY
done
