#!/bin/bash
# round 2, GPU call B: reference goldens (fixed wrapper), full-size leg B with loss-fragile mask, SQ counters of both K8 variants
O=gpurun_out/r2b; mkdir -p $O; export TMPDIR=/tmp PYTHONUNBUFFERED=1
R=$GRAFT_REPO_ROOT
( time timeout 400 python tests/golden/make_reference_cuda_golden.py ) > $O/refgolden.log 2>&1; echo "refgolden rc=$?" >> $O/summary.txt
( timeout 1500 python -m pytest tests/test_gpu_fused_fullsize.py -q -m gpu -s ) > $O/fullsize.log 2>&1; echo "fullsize rc=$?" >> $O/summary.txt
for v in scan cell; do
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_ANY" \
           "SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT"; do
  i=$((i+1))
  ( cd /tmp && GHR_K8=$v timeout 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pmc$v$i -o p -- python $R/tools/kbench.py cfg3 5 ) > $O/pmc_$v$i.log 2>&1
  python - <<PY >> $O/pmc_sq.log
import csv,glob,collections
f=glob.glob('/tmp/pmc$v$i/**/*counter_collection.csv',recursive=True)
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for fn in f:
    for r in csv.DictReader(open(fn)):
        k=r['Kernel_Name'].split('(')[0]
        if 'k_render_bwd' in k: acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k in acc:
    print('PMC $v', k, ' '.join('%s=%.4g' % (c, sum(v)/len(v)) for c,v in acc[k].items()))
PY
done; done
cat $O/summary.txt; cat $O/pmc_sq.log; tail -5 $O/refgolden.log; grep -n "AssertionError:\|fullsize cfg\|passed\|failed" $O/fullsize.log
