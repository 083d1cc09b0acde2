#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=$PWD/gpurun_out/r06stp; mkdir -p $O; export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 600 python tools/strand_torch_profile.py > $O/strand_torch_profile.txt 2>&1
tail -5 $O/strand_torch_profile.txt
