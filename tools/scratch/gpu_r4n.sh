#!/bin/bash
O=gpurun_out/r4n; mkdir -p $O; export TMPDIR=/tmp PYTHONUNBUFFERED=1
V=$GRAFT_REPO_ROOT/gaussianhaircut_amd/csrc/variants
bash tools/stepbench.sh "" $V/libghr_fakepix.so $V/libghr_noatom.so "" $V/libghr_fakepix.so $V/libghr_noatom.so
