#!/bin/bash
# SQ counters of the K8 variants over tools/kbench.py: tools/gpu/sqpmc.sh <outfile> <k8> [cfg]
cd ${GRAFT_REPO_ROOT:-/root/repo}
out=$1; k8=$2; cfg=${3:-cfg3}
export TMPDIR=/tmp
R=$PWD
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_LDS" "SQ_INSTS_SALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM"; do
  i=$((i+1)); rm -rf /tmp/sq$i
  ( cd /tmp && GHR_K8=$k8 timeout 120 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/sq$i -o p -- python $R/tools/kbench.py $cfg 5 ) > /tmp/sq$i.log 2>&1
  python - <<PY >> $out
import csv,glob,collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for fn in glob.glob('/tmp/sq$i/**/*counter_collection.csv',recursive=True):
    for r in csv.DictReader(open(fn)):
        k=r['Kernel_Name'].split('(')[0].replace('ghr::','')
        if k.startswith('k_render'): acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k in sorted(acc):
    print('PMC [$k8 $cfg]', k, ' '.join('%s=%.4g' % (c, sum(v)/len(v)) for c,v in acc[k].items()))
PY
done
