#!/bin/bash
# full GPU suite + smoke + bench at HEAD, then the sort-cliff sweep
O=gpurun_out/r2p; mkdir -p $O; export TMPDIR=/tmp PYTHONUNBUFFERED=1
R=$GRAFT_REPO_ROOT
( timeout 1700 python -m pytest tests -q -m gpu --timeout 1500 -x ) > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -5 $O/pytest.log
( timeout 300 python __graft_entry__.py smoke ) > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log; tail -2 $O/smoke.log
( timeout 600 python bench.py --steps 20 --warmup 5 ) > $O/bench.json 2> $O/bench.err; tail -c 3000 $O/bench.json
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/sc -- python $R/tools/sort_cliff.py ) > $O/cliff_run.log 2>&1
python tools/sort_cliff.py --parse /tmp/sc > $O/sort_cliff.txt 2>&1; cat $O/sort_cliff.txt
