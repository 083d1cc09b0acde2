#!/bin/bash
# Kernel-experiment build:  tools/build_variant.sh <name> [-p <patch> ...] [-DFLAG ...] -> build/variants/libghr_<name>.so
# (select with GHR_LIB_PATH; *.so is git-ignored but travels to the GPU box).
# Rejected kernel forms and their ablation knobs do not live in the product headers: they are patches under
# tools/experiments/ (e.g. r03_experiment_knobs_of_the_render_loss_adam_kernels.patch brings back -DGHR_B3_NOARITH,
# -DGHR_B3_NOATOM, -DGHR_K7_HALVES, ...).  -p applies a patch to a scratch copy of csrc/ + include/ before compiling.
set -e -o pipefail
R=$(cd "$(dirname "$0")/.." && pwd)
name=$1; shift
S=$R/build/variants/src_$name
rm -rf $S; mkdir -p $S/gaussianhaircut_amd $R/build/variants
cp -r $R/gaussianhaircut_amd/csrc $S/gaussianhaircut_amd/csrc
cp -r $R/include $S/include
rm -f $S/gaussianhaircut_amd/csrc/*.so
flags=()
while [ $# -gt 0 ]; do
  if [ "$1" = "-p" ]; then (cd $S && patch -s -p1 < "$(cd $R && realpath "$2")"); shift 2; else flags+=("$1"); shift; fi
done
# (a failed compile must not leave a stale libghr_<name>.so behind to be measured as if it were the new variant)
out=$R/build/variants/libghr_$name.so
rm -f $out
log=$(/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -munsafe-fp-atomics -fPIC -shared \
  -I$S/include -I$S/gaussianhaircut_amd/csrc "${flags[@]}" $S/gaussianhaircut_amd/csrc/ghr_capi.hip -o $out 2>&1) || rc=$?
echo "$log" | grep -E "error|spill" || true
rm -rf $S
if [ -n "${rc:-}" ] || [ ! -s $out ]; then echo "build_variant: hipcc FAILED for $name (rc=${rc:-0})" >&2; exit 1; fi
echo built $out
