#!/bin/bash
# rocprofv3 passes on the GPU box: kernel trace + stats, then FETCH_SIZE and WRITE_SIZE in their own passes.
mkdir -p gpurun_out
export TMPDIR=/tmp
TAG=${1:-r01}
R=$GRAFT_REPO_ROOT
B="python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-op-only --streams 1"  # one stream: per-kernel durations without co-running views
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_kt -o kt -- $B ) > gpurun_out/rocprof_kt.log 2>&1; echo "kt rc=$?"
( cd /tmp && timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/prof_fetch -o f -- $B ) > gpurun_out/rocprof_fetch.log 2>&1; echo "fetch rc=$?"
( cd /tmp && timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/prof_write -o w -- $B ) > gpurun_out/rocprof_write.log 2>&1; echo "write rc=$?"
mkdir -p /tmp/prof_all && cp -r /tmp/prof_kt /tmp/prof_fetch /tmp/prof_write /tmp/prof_all/ 2>/dev/null
find /tmp/prof_all -name "*.csv" | head -20
python tools/parse_prof.py /tmp/prof_all gpurun_out/profiles $TAG
