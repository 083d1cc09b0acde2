#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
R=$PWD; P=$PWD/gpurun_out/r06s; mkdir -p $P; export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k recycles 2>&1 | tail -4 | tee $P/pytest.log
cat > /tmp/oponly.py <<PY
import sys, json, torch
sys.path.insert(0, "$R")
import bench
for cfg in ("cfg3", "cfg2"):
    print("OPONLY", cfg, json.dumps({k: v for k, v in bench.op_only_bench(torch.device("cuda:0"), cfg, iters=40).items() if k in ("fwd_ms", "bwd_ms", "bwd_ms_host_sync_every_iter", "fwd_ms_host_sync_every_iter")}))
PY
