#!/bin/bash
# SQ counters for every kernel of the bench step (GPU box tool); separate --pmc passes, each time-boxed
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_ANY" \
           "SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD" \
           "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VMEM SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_VMEM SQ_INSTS_FLAT SQ_LEVEL_WAVES SQ_ACCUM_PREV_HIRES SQ_CYCLES"; do
  i=$((i+1))
  ( cd /tmp && timeout 240 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pa$i -o p -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-op-only --streams 1 ) > gpurun_out/pa$i.log 2>&1
  python - <<PY
import csv,glob,collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for fn in glob.glob('/tmp/pa$i/**/*counter_collection.csv',recursive=True):
    for r in csv.DictReader(open(fn)):
        k=r['Kernel_Name'].split('(')[0].replace('ghr::','')
        if k.startswith('k_'): acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k in sorted(acc):
    print('PMC', k, ' '.join('%s=%.4g' % (c, sum(v)/len(v)) for c,v in acc[k].items()))
PY
done | tee gpurun_out/pmc_all.log
