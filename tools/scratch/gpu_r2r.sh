#!/bin/bash
O=gpurun_out/r2r; mkdir -p $O; export TMPDIR=/tmp PYTHONUNBUFFERED=1
R=$GRAFT_REPO_ROOT
export GHR_PROF_NAMES="zero+barrier,cell fetch/pixels/idle,mask list,staging,chunks,-"
for c in cfg3 cfg2; do
( GHR_K8=cells GHR_LIB_PATH=$R/gaussianhaircut_amd/csrc/variants/libghr_prof.so timeout 300 python tools/kbench.py $c 10 ) 2>&1 | grep -E "KBENCH|PROF|rror" >> $O/prof.log
done
cat $O/prof.log
v=cells
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_ANY" \
           "SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_FLAT SQ_LEVEL_WAVES"; do
  i=$((i+1))
  ( cd /tmp && GHR_K8=$v timeout 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pmc$v$i -o p -- python $R/tools/kbench.py cfg3 5 ) > $O/pmc_$v$i.log 2>&1
  python - <<PY >> $O/pmc_sq.log
import csv,glob,collections
f=glob.glob('/tmp/pmc$v$i/**/*counter_collection.csv',recursive=True)
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for fn in f:
    for r in csv.DictReader(open(fn)):
        k=r['Kernel_Name'].split('(')[0]
        if 'k_render' in k: acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k in acc:
    print('PMC $v', k, ' '.join('%s=%.4g' % (c, sum(v)/len(v)) for c,v in acc[k].items()))
PY
done
cat $O/pmc_sq.log
