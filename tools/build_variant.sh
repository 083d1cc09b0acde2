#!/bin/bash
# Kernel-experiment build: tools/build_variant.sh <name> [-DFLAG ...] -> build/variants/libghr_<name>.so
# (select with GHR_LIB_PATH; *.so is git-ignored but travels to the GPU box)
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
name=$1; shift
mkdir -p $R/build/variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -munsafe-fp-atomics -fPIC -shared \
  -I$R/include -I$R/gaussianhaircut_amd/csrc "$@" $R/gaussianhaircut_amd/csrc/ghr_capi.hip \
  -o $R/build/variants/libghr_$name.so
echo built $R/build/variants/libghr_$name.so
