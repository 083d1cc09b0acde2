#!/bin/bash
# SQ counters of the kernels whose name starts with <prefix>, over the single-view step of bench.py (one stream):
#   tools/gpu/sqpmc_step.sh <outfile> <prefix> [ENV=val ...]
cd ${GRAFT_REPO_ROOT:-/root/repo}
out=$1; pre=$2; shift; shift
export TMPDIR=/tmp
R=$PWD
B="python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-op-only --streams 1 --shard-views 0"
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_LDS" "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM SQ_INST_CYCLES_SALU" "SQ_IFETCH SQ_IFETCH_LEVEL SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT" "TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TA_TA_BUSY_sum GRBM_GUI_ACTIVE SQ_INSTS_VALU"; do
  i=$((i+1)); rm -rf /tmp/sqs$i
  ( cd /tmp && env "$@" timeout 240 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/sqs$i -o p -- $B ) > /tmp/sqs$i.log 2>&1
  python - <<PY >> $out
import csv,glob,collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for fn in glob.glob('/tmp/sqs$i/**/*counter_collection.csv',recursive=True):
    for r in csv.DictReader(open(fn)):
        k=r['Kernel_Name'].split('(')[0].replace('ghr::','')
        if k.startswith('$pre'): acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k in sorted(acc):
    print('PMC [$*]', k, ' '.join('%s=%.4g' % (c, sum(v)/len(v)) for c,v in acc[k].items()))
if not acc: print('PMC set $i: nothing', open('/tmp/sqs$i.log').read()[-600:])
PY
done
