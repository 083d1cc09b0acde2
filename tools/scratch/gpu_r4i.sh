#!/bin/bash
O=gpurun_out/r4i; mkdir -p $O; export TMPDIR=/tmp PYTHONUNBUFFERED=1
V=$GRAFT_REPO_ROOT/gaussianhaircut_amd/csrc/variants
for rep in 1 2; do for lib in "" $V/libghr_nozero.so; do
( GHR_LIB_PATH=$lib timeout 120 python tools/kbench.py cfg3 30 ) 2>&1 | grep -E "KBENCH|rror"
done; done
bash tools/stepbench.sh "" $V/libghr_nozero.so "" $V/libghr_nozero.so
