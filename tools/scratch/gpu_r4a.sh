#!/bin/bash
O=gpurun_out/r4a; mkdir -p $O; export TMPDIR=/tmp PYTHONUNBUFFERED=1
R=$GRAFT_REPO_ROOT
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/sc3 -- python $R/tools/sort_cliff.py ) > $O/cliff.log 2>&1
python tools/sort_cliff.py --parse /tmp/sc3 | tee $O/sort_cliff.txt
bash tools/stepbench.sh "" ""
