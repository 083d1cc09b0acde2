#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r06sh; mkdir -p $O; export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 1800 python -m pytest tests/test_gpu_dist_shared.py -m gpu -x -q 2>&1 | tail -15 | tee $O/pytest_dist.log
timeout 900 python -m pytest tests/test_gpu_fused.py -m gpu -x -q -k "fold_their_sh or per_view_factors" 2>&1 | tail -5 | tee $O/pytest_fold.log
for rep in 1 2; do
MASTER_PORT=29587 python tools/shardstep.py 1 40 2>&1 | grep "SHARDSTEP views" | tee -a $O/shardstep_flag_in_rows.txt
done
