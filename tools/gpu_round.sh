#!/bin/bash
# One gpurun call: smoke, stage-by-stage diagnostic, gpu tests, bench, rocprof kernel trace.  Everything is logged
# under gpurun_out/ and every step is time-boxed so a hang cannot become a strike.
mkdir -p gpurun_out
export TMPDIR=/tmp
rocminfo | grep -E "Marketing Name|gfx" | head -4 > gpurun_out/gpu.txt 2>&1
( time timeout 300 python __graft_entry__.py smoke ) > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/summary.txt

( time timeout 900 python -m pytest tests -m gpu -q -x --timeout 600 ) > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/summary.txt
( time timeout 600 python bench.py --steps 10 --warmup 3 ) > gpurun_out/bench.log 2>&1; echo "bench rc=$?" >> gpurun_out/summary.txt
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof -o r01 -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-op-only ) > gpurun_out/rocprof.log 2>&1; echo "rocprof rc=$?" >> gpurun_out/summary.txt
# keep only small rocprof summaries (gpurun_out is capped at 64 MiB)
find gpurun_out/prof -type f ! -name "*stats*.csv" ! -name "*agent_info*.csv" -delete 2>/dev/null
( time timeout 600 python tools/torch_profile.py cfg3 ) > gpurun_out/torch_profile.log 2>&1; echo "torchprof rc=$?" >> gpurun_out/summary.txt
cat gpurun_out/summary.txt
tail -5 gpurun_out/smoke.log; tail -60 gpurun_out/diag.log; tail -30 gpurun_out/pytest_gpu.log; tail -5 gpurun_out/bench.log
