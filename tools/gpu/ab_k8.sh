#!/bin/bash
# kbench A/B of library variants without the parity run: tools/gpu/ab_k8.sh <outdir> <cfg> <variant ...>
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/$1; cfg=$2; shift; shift; mkdir -p $O
export TMPDIR=/tmp
for v in "$@"; do
  lib=""; [ "$v" != product ] && lib=$PWD/build/variants/libghr_$v.so
  GHR_LIB_PATH=$lib timeout 120 python tools/kbench.py $cfg 30 2>&1 | grep "KBENCH" | sed "s/^/[$v] /" | tee -a $O/kbench.log
done
