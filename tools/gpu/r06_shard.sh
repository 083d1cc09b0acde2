#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r06sh; mkdir -p $O; export TMPDIR=/tmp PYTHONUNBUFFERED=1
rm -f $O/shardstep.log
for rep in 1 2; do
for V in 4 1; do
  GHR_FACTORED_SH_REDUCE=0 MASTER_PORT=29581 python tools/shardstep.py $V 20 2>&1 | grep SHARDSTEP | tee -a $O/shardstep.log
  GHR_SH_MAX_VIEWS=0 MASTER_PORT=29582 python tools/shardstep.py $V 20 2>&1 | grep SHARDSTEP | tee -a $O/shardstep.log
  GHR_SH_MAX_VIEWS=16 MASTER_PORT=29583 python tools/shardstep.py $V 20 2>&1 | grep SHARDSTEP | tee -a $O/shardstep.log
done
done
