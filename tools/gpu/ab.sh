#!/bin/bash
# A/B of library variants on the GPU box: tools/gpu/ab.sh <outdir> <variant ...>   ("product" = the in-tree build)
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/$1; shift; mkdir -p $O; rm -f $O/*
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_reference_cuda_golden.py tests/test_gpu_fused.py -m gpu -x -q 2>&1 | tail -5 > $O/pytest.log
tail -2 $O/pytest.log
for v in "$@"; do
  lib=""; [ "$v" != product ] && lib=$PWD/build/variants/libghr_$v.so
  for cfg in cfg3 cfg2 cfg5; do
    GHR_LIB_PATH=$lib timeout 300 python tools/kbench.py $cfg 30 2>&1 | grep "KBENCH\|PROF" | sed "s/^/[$v] /" >> $O/kbench.log
  done
done
cat $O/kbench.log
