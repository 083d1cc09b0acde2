#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r06hf; mkdir -p $O; export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_gpu_fused.py -m gpu -x -q -k "strand_stage_direct or strand_training_step_learns" 2>&1 | grep -v "amdgpu.ids" | tail -30 | tee $O/pytest.log
