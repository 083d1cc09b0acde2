#!/bin/bash
O=gpurun_out/r2l; mkdir -p $O; export TMPDIR=/tmp PYTHONUNBUFFERED=1
R=$GRAFT_REPO_ROOT
( timeout 2400 python -m pytest tests -q -m gpu -x --timeout 1500 ) > $O/gputests.log 2>&1; echo "gputests rc=$?" >> $O/summary.txt
( timeout 900 python bench.py --steps 20 --warmup 5 ) > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" >> $O/summary.txt
cat $O/summary.txt; tail -15 $O/gputests.log; cat $O/bench.json; tail -5 $O/bench.err
