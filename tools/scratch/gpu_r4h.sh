#!/bin/bash
O=gpurun_out/r4h; mkdir -p $O; export TMPDIR=/tmp PYTHONUNBUFFERED=1
bash tools/gpu_profile_r02.sh r02d > $O/profile.log 2>&1
grep "^KT\|rc=\|^PMC {" $O/profile.log | head -24
