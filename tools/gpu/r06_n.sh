#!/bin/bash
# op-only backward (GaussianRasterizer autograd op, cfg3): round 5's kernels against round 6's, same box
cd ${GRAFT_REPO_ROOT:-/root/repo}
R=$PWD; P=$PWD/gpurun_out/r06n; mkdir -p $P; export TMPDIR=/tmp PYTHONUNBUFFERED=1
cat > /tmp/oponly.py <<PY
import sys, json, torch
sys.path.insert(0, "$R")
import bench
for cfg in ("cfg3", "cfg2"):
    print("OPONLY", cfg, json.dumps({k: v for k, v in bench.op_only_bench(torch.device("cuda:0"), cfg, iters=40).items() if k in ("fwd_ms", "bwd_ms", "bwd_ms_host_sync_every_iter", "fwd_ms_host_sync_every_iter")}))
PY
for v in product r5abi17 product r5abi17; do
  lib=""; [ "$v" != product ] && lib=$PWD/build/variants/libghr_$v.so
  GHR_LIB_PATH=$lib python /tmp/oponly.py 2>&1 | grep OPONLY | sed "s/^/[$v] /" | tee -a $P/oponly.log
done
