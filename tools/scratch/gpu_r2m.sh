#!/bin/bash
O=gpurun_out/r2m; mkdir -p $O; export TMPDIR=/tmp PYTHONUNBUFFERED=1
( timeout 2400 python -m pytest tests -q -m gpu --timeout 1500 ) > $O/gputests.log 2>&1; echo "gputests rc=$?" >> $O/summary.txt
bash tools/gpu_profile_r02.sh r02a > $O/profile.log 2>&1
cat $O/summary.txt; tail -12 $O/gputests.log; tail -40 $O/profile.log
