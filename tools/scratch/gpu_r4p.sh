#!/bin/bash
# full GPU suite + smoke + bench with the cell-list K8 as default
O=gpurun_out/r4p; mkdir -p $O; export TMPDIR=/tmp PYTHONUNBUFFERED=1
( timeout 1700 python -m pytest tests -q -m gpu --timeout 1500 ) > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -6 $O/pytest.log
( timeout 300 python __graft_entry__.py smoke ) > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log; tail -2 $O/smoke.log
( timeout 600 python bench.py --steps 20 --warmup 5 ) > $O/bench.json 2> $O/bench.err; tail -c 2500 $O/bench.json
for v in cells cell; do
( GHR_K8=$v timeout 120 python tools/kbench.py cfg3 30 ) 2>&1 | grep -E "KBENCH|rror" | sed "s/^/$v /" >> $O/kbench.log
done; cat $O/kbench.log
