#!/bin/bash
O=gpurun_out/r3x; mkdir -p $O; export TMPDIR=/tmp PYTHONUNBUFFERED=1
R=$GRAFT_REPO_ROOT
for v in cells cell; do
( GHR_K8=$v timeout 300 python tools/bench_hair.py ) 2>&1 | grep -v Warning | tail -3 | sed "s/^/$v: /"
done
for v in cells cell; do
( cd /tmp && GHR_K8=$v timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/sc_$v -- python $R/tools/sort_cliff.py ) > $O/cliff_$v.log 2>&1
echo "== $v"; python tools/sort_cliff.py --parse /tmp/sc_$v
done
