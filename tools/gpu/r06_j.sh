#!/bin/bash
# round 6: K8 with an LDS line table (ds_add_f32) for tiles of <= 192 instances, built as a variant library: parity, then A/B
cd ${GRAFT_REPO_ROOT:-/root/repo}
P=$PWD/gpurun_out/r06j; mkdir -p $P; export TMPDIR=/tmp PYTHONUNBUFFERED=1
V=$PWD/build/variants/libghr_table.so
GHR_LIB_PATH=$V GHR_K8_TABLE=1 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fused_fullsize.py tests/test_gpu_reference_live.py -m gpu -q -x 2>&1 | tail -12 | tee $P/parity_table_on.log
for cfg in cfg2 cfg3 cfg1; do
  timeout 120 python tools/kbench.py $cfg 30 2>&1 | grep KBENCH | sed "s/^/[product] /" | tee -a $P/kbench.log
  GHR_LIB_PATH=$V GHR_K8_TABLE=0 timeout 120 python tools/kbench.py $cfg 30 2>&1 | grep KBENCH | sed "s/^/[variant, table off] /" | tee -a $P/kbench.log
  GHR_LIB_PATH=$V GHR_K8_TABLE=1 timeout 120 python tools/kbench.py $cfg 30 2>&1 | grep KBENCH | sed "s/^/[variant, table ON: tiles <= 192] /" | tee -a $P/kbench.log
  timeout 120 python tools/kbench.py $cfg 30 2>&1 | grep KBENCH | sed "s/^/[product] /" | tee -a $P/kbench.log
done
