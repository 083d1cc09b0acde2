#!/bin/bash
O=gpurun_out/r3v; mkdir -p $O; export TMPDIR=/tmp PYTHONUNBUFFERED=1
V=$GRAFT_REPO_ROOT/gaussianhaircut_amd/csrc/variants
for rep in 1 2; do
for lib in "" $V/libghr_static.so $V/libghr_skipz.so $V/libghr_both.so; do
( GHR_LIB_PATH=$lib timeout 120 python tools/kbench.py cfg3 30 ) 2>&1 | grep -E "KBENCH|rror"
done; done
bash tools/stepbench.sh "" $V/libghr_static.so $V/libghr_skipz.so $V/libghr_both.so ""
( GHR_LIB_PATH=$V/libghr_both.so timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x --timeout 500 -k "not deterministic" ) 2>&1 | tail -1
