#!/bin/bash
# round 5, call n: raw parameters / record / rect in the first round trip of k_project_bwd, slab by LDS-DMA, branch-free gather
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/n; export PYTHONUNBUFFERED=1
timeout 600 python -m pytest tests/test_gpu_fused.py tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/n/tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/n/tests.log
V=$PWD/build/variants
bash tools/gpu/kt.sh n_head1 GHR_LIB_PATH=$V/libghr_head.so | grep "k_project\|k_geom\|ms_per"
bash tools/gpu/kt.sh n_new1 | grep "k_project\|k_geom\|ms_per"
bash tools/gpu/kt.sh n_head2 GHR_LIB_PATH=$V/libghr_head.so | grep "k_project\|k_geom\|ms_per"
bash tools/gpu/kt.sh n_new2 | grep "k_project\|k_geom\|ms_per"
