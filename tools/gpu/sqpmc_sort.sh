#!/bin/bash
# SQ counters of the binning kernels over tools/kbench.py: tools/gpu/sqpmc_sort.sh <outfile> [cfg] [ENV=..]
cd ${GRAFT_REPO_ROOT:-/root/repo}
out=$1; cfg=${2:-cfg3}; shift; shift
export TMPDIR=/tmp
R=$PWD
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_LDS" "SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA"; do
  i=$((i+1)); rm -rf /tmp/sqs$i
  ( cd /tmp && env "$@" timeout 120 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/sqs$i -o p -- python $R/tools/kbench.py $cfg 5 ) > /tmp/sqs$i.log 2>&1
  python - <<PY >> $out
import csv,glob,collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for fn in glob.glob('/tmp/sqs$i/**/*counter_collection.csv',recursive=True):
    for r in csv.DictReader(open(fn)):
        k=r['Kernel_Name'].split('(')[0].replace('ghr::','')
        if k.startswith('k_tile_sort'): acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k in sorted(acc):
    print('PMC [$cfg $*]', k, ' '.join('%s=%.4g' % (c, sum(v)/len(v)) for c,v in acc[k].items()))
PY
done
