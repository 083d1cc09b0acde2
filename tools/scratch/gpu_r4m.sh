#!/bin/bash
O=gpurun_out/r4m; mkdir -p $O; export TMPDIR=/tmp PYTHONUNBUFFERED=1
( timeout 1700 python -m pytest tests -q -m gpu --timeout 1500 ) > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -2 $O/pytest.log
( timeout 300 python __graft_entry__.py smoke ) > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log; tail -2 $O/smoke.log
( timeout 600 python bench.py --steps 20 --warmup 5 ) > $O/bench.json 2> $O/bench.err
bash tools/gpu_profile_r02.sh r02e > $O/profile.log 2>&1
grep "^KT\|rc=" $O/profile.log | head -14
for c in cfg3 cfg2 cfg5; do ( timeout 120 python tools/kbench.py $c 30 ) 2>&1 | grep -E "KBENCH|rror"; done
