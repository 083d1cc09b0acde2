#!/bin/bash
# round 6, second half: the whole GPU suite, then the profile pass
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r06z; mkdir -p $O; export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 | tee $O/pytest.log
bash tools/gpu/profile_r06.sh r06
