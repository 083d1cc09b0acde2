#!/bin/bash
# round 6: K8 with the geometric reductions by v_permlane swaps + two placing MFMAs (-DGHR_B3_SWAP_REDUCE): parity, kbench A/B
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r06k8; mkdir -p $O; export TMPDIR=/tmp
V=$PWD/build/variants/libghr_k8swap.so
GHR_LIB_PATH=$V timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_reference_cuda_golden.py tests/test_gpu_reference_live.py -m gpu -x -q 2>&1 | tail -6 | tee $O/pytest_variant.log
rm -f $O/kbench.log
for rep in 1 2 3; do
for v in product k8swap; do
  lib=""; [ "$v" != product ] && lib=$V
  for cfg in cfg3 cfg2 cfg5; do
    GHR_LIB_PATH=$lib timeout 300 python tools/kbench.py $cfg 30 2>&1 | grep "KBENCH" | sed "s/^/[$v] /" >> $O/kbench.log
  done
done
done
cat $O/kbench.log
