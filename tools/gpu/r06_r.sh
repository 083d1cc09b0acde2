#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
P=$PWD/gpurun_out/r06r; mkdir -p $P; export TMPDIR=/tmp PYTHONUNBUFFERED=1
python tools/host_profile.py 400 2>&1 | grep HOST | head -14 | tee $P/host_profile.txt
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -4 | tee $P/pytest_all.log
python tools/camstep.py plain 100 2>&1 | grep CAMSTEP | tee $P/camstep.log
