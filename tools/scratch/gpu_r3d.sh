#!/bin/bash
O=gpurun_out/r3d; mkdir -p $O; export TMPDIR=/tmp PYTHONUNBUFFERED=1
R=$GRAFT_REPO_ROOT
bash tools/gpu_profile_r02.sh r02b > $O/profile.log 2>&1
tail -30 $O/profile.log
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_ANY" \
           "SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT"; do
  i=$((i+1))
  ( cd /tmp && timeout 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pmcs$i -o p -- python $R/tools/kbench.py cfg3 5 ) > $O/pmc_$i.log 2>&1
  python - <<PY >> $O/pmc_sq.log
import csv,glob,collections
f=glob.glob('/tmp/pmcs$i/**/*counter_collection.csv',recursive=True)
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for fn in f:
    for r in csv.DictReader(open(fn)):
        k=r['Kernel_Name'].split('(')[0]
        if 'k_render' in k: acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k in acc:
    print('PMC', k, ' '.join('%s=%.4g' % (c, sum(v)/len(v)) for c,v in acc[k].items()))
PY
done
cat $O/pmc_sq.log
