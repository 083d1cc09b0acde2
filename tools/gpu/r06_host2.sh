#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r06sh; mkdir -p $O; export TMPDIR=/tmp PYTHONUNBUFFERED=1
for rep in 1 2 3; do
GHR_SHARD_WITH_GATHERED_VIEWS=1 MASTER_PORT=29585 python tools/shardstep.py 1 40 2>&1 | grep "SHARDSTEP views" | sed 's/^/shard+gather /' | tee -a $O/shardstep_gather_shard_ab.txt
GHR_SHARD_WITH_GATHERED_VIEWS=0 MASTER_PORT=29586 python tools/shardstep.py 1 40 2>&1 | grep "SHARDSTEP views" | sed 's/^/allreduce+gather /' | tee -a $O/shardstep_gather_shard_ab.txt
done
