#!/bin/bash
O=gpurun_out/r3w; mkdir -p $O; export TMPDIR=/tmp PYTHONUNBUFFERED=1
bash tools/gpu_profile_r02.sh r02c > $O/profile.log 2>&1
grep "^KT\|rc=" $O/profile.log | head -30
