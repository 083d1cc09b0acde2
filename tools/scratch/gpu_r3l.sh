#!/bin/bash
O=gpurun_out/r3l; mkdir -p $O; export TMPDIR=/tmp PYTHONUNBUFFERED=1
V=$GRAFT_REPO_ROOT/gaussianhaircut_amd/csrc/variants
( bash tools/stepbench.sh "" $V/libghr_warm1.so $V/libghr_warm2.so $V/libghr_warm3.so $V/libghr_w4.so; GHR_K8=cell bash tools/stepbench.sh ""; bash tools/stepbench.sh "" ) 2>&1 | tee $O/step.log
