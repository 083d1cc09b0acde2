#!/bin/bash
O=gpurun_out/r4q; mkdir -p $O; export TMPDIR=/tmp PYTHONUNBUFFERED=1
V=$GRAFT_REPO_ROOT/gaussianhaircut_amd/csrc/variants
bash tools/stepbench.sh "" $V/libghr_pixfirst.so "" $V/libghr_pixfirst.so
( GHR_LIB_PATH=$V/libghr_pixfirst.so timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x --timeout 500 ) 2>&1 | tail -1
