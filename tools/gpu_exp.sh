#!/bin/bash
# kernel experiments: time each variant library, then (PMC=1) SQ counters on the default library
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for so in $R/gaussianhaircut_amd/csrc/variants/*.so; do
  GHR_LIB_PATH=$so timeout 90 python tools/kbench.py ${CFG:-cfg3} 20 2>&1 | grep -E "KBENCH|Error|error"
done | tee gpurun_out/kbench.log
if [ -n "$KT" ]; then
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -o kt -- python $R/tools/kbench.py ${CFG:-cfg3} 10 ) > gpurun_out/kt.log 2>&1
  python - <<PY | tee gpurun_out/kt_stats.log
import csv,glob
for f in glob.glob('/tmp/kt/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'ghr::' in r['Name']:
            print('KT %-18s calls %3s avg %9.1f us min %9.1f us' % (r['Name'].split('(')[0].replace('ghr::',''), r['Calls'], float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3))
PY
fi
if [ -n "$PMC" ]; then
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_ANY" \
           "SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR" \
           "GRBM_GUI_ACTIVE SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_LDS_ATOMIC_RETURN SQ_INSTS_FLAT"; do
  i=$((i+1))
  ( cd /tmp && timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pmc$i -o p -- python $R/tools/kbench.py ${CFG:-cfg3} 5 ) > gpurun_out/pmc$i.log 2>&1
  python - <<PY
import csv,glob,collections
f=glob.glob('/tmp/pmc$i/**/*counter_collection.csv',recursive=True)
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for fn in f:
    for r in csv.DictReader(open(fn)):
        k=r['Kernel_Name'].split('(')[0]
        if 'k_render' in k: acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k in acc:
    for c,v in acc[k].items(): print('PMC',k,c,sum(v)/len(v),len(v))
PY
done | tee gpurun_out/pmc_sq.log
fi
