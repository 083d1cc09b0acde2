#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/y
timeout 600 python -m pytest tests/test_gpu_fused.py -x -q -m gpu -k "test_fused_render_matches_generic_path" 2>&1 | tail -15
