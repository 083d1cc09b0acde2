#!/bin/bash
# round 6: the densification event's re-lay as one HIP pass (ghr_adam_relay_rows): the bit test, then the event timed both ways
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r06relay; mkdir -p $O; export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_gpu_fused.py tests/test_gpu_loss_adam.py -m gpu -x -q -k "densif or onepass or surgery" 2>&1 | tail -4 | tee $O/pytest.log
GHR_DENSIFY_RELAY_KERNEL=0 timeout 600 python tools/densify_bench.py cfg3 2>&1 | grep DENSIFY | sed 's/^/[one-pass, PyTorch re-lay]  /' | tee $O/densify.txt
GHR_DENSIFY_RELAY_KERNEL=1 timeout 600 python tools/densify_bench.py cfg3 2>&1 | grep DENSIFY | sed 's/^/[one-pass, HIP re-lay kernel]  /' | tee -a $O/densify.txt
