#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r3o; mkdir -p $O; rm -f $O/*
export TMPDIR=/tmp
python -m pytest tests/test_gpu_parity.py tests/test_reference_cuda_golden.py tests/test_gpu_fused.py -m gpu -x -q 2>&1 | tail -2
GHR_LIB_PATH=$PWD/build/variants/libghr_q1280.so python -m pytest tests/test_gpu_parity.py tests/test_reference_cuda_golden.py tests/test_gpu_fused.py -m gpu -x -q 2>&1 | tail -2
for v in product q1280; do
  lib=""; [ "$v" != product ] && lib=$PWD/build/variants/libghr_$v.so
  for ord in 3 7; do for cfg in cfg3 cfg2 cfg5; do
    GHR_TILE_ORDER=$ord GHR_LIB_PATH=$lib timeout 300 python tools/kbench.py $cfg 30 2>&1 | grep "KBENCH" | sed "s/^/[$v order $ord] /" >> $O/kbench.log
  done; done
done
cat $O/kbench.log
