#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r06mini; mkdir -p $O; export TMPDIR=/tmp PYTHONUNBUFFERED=1
python tools/strandstep.py 100 2>&1 | grep STRAND | tee $O/strand_100.txt
