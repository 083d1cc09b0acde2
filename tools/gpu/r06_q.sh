#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
P=$PWD/gpurun_out/r06q; mkdir -p $P; export TMPDIR=/tmp PYTHONUNBUFFERED=1
python tools/host_profile.py 400 2>&1 | grep HOST | tee $P/host_profile.txt
