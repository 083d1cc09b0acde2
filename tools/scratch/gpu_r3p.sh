#!/bin/bash
O=gpurun_out/r3p; mkdir -p $O; export TMPDIR=/tmp PYTHONUNBUFFERED=1
R=$GRAFT_REPO_ROOT
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/tl -o tl -- python $R/bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-op-only --streams 1 --shard-views 0 ) > $O/tl.log 2>&1
python tools/timeline.py /tmp/tl -3 > $O/timeline.txt 2>&1; cat $O/timeline.txt
