#!/bin/bash
O=gpurun_out/r2u; mkdir -p $O; export TMPDIR=/tmp PYTHONUNBUFFERED=1
R=$GRAFT_REPO_ROOT
V=$R/gaussianhaircut_amd/csrc/variants
export GHR_PROF_NAMES="-,cell fetch/pixels/idle,mask list,staging,chunks,-"
export GHR_K8=cells
for lib in "" $V/libghr_w6.so $V/libghr_w5na.so $V/libghr_prof5.so; do
for c in cfg3 cfg2; do
( GHR_LIB_PATH=$lib timeout 300 python tools/kbench.py $c 20 ) 2>&1 | grep -E "KBENCH|PROF|rror" >> $O/prof.log
done; done
cat $O/prof.log
( timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_reference_cuda_golden.py -q -m gpu --timeout 1000 -x ) > $O/pytest_cells.log 2>&1; echo "rc=$?" >> $O/pytest_cells.log
tail -4 $O/pytest_cells.log
