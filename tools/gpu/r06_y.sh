#!/bin/bash
# round 6: K7 variant A/B (the library named in V): parity with the variant library, then kbench
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r06k7; mkdir -p $O; export TMPDIR=/tmp
V=$PWD/build/variants/libghr_k7words.so
GHR_LIB_PATH=$V timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_reference_cuda_golden.py tests/test_gpu_reference_live.py tests/test_gpu_fused.py -m gpu -x -q 2>&1 | tail -12 | tee $O/pytest_variant.log
rm -f $O/kbench.log
for rep in 1 2 3; do
for v in product k7words; do
  lib=""; [ "$v" != product ] && lib=$V
  for cfg in cfg3 cfg2 cfg5; do
    GHR_LIB_PATH=$lib timeout 300 python tools/kbench.py $cfg 30 2>&1 | grep "KBENCH" | sed "s/^/[$v] /" >> $O/kbench.log
  done
done
done
cat $O/kbench.log
