#!/bin/bash
# round 2, GPU call A: reference goldens, wave primitives, parity of the scan K8, timing of both K8 variants, new full-size tests
O=gpurun_out/r2a; mkdir -p $O
export PYTHONUNBUFFERED=1
( time timeout 400 python tests/golden/make_reference_cuda_golden.py ) > $O/refgolden.log 2>&1; echo "refgolden rc=$?" >> $O/summary.txt
( timeout 300 python -m pytest tests/test_gpu_wave_primitives.py -q -m gpu ) > $O/wave.log 2>&1; echo "wave rc=$?" >> $O/summary.txt
( timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu ) > $O/parity_scan.log 2>&1; echo "parity_scan rc=$?" >> $O/summary.txt
( GHR_K8=cell timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "tiny or cfg1 or ragged" ) > $O/parity_cell.log 2>&1; echo "parity_cell rc=$?" >> $O/summary.txt
for v in scan cell; do for c in cfg3 cfg2; do
  ( GHR_K8=$v timeout 200 python tools/kbench.py $c 20 ) 2>&1 | grep -E "KBENCH|rror" >> $O/kbench.log; done; done
( timeout 1500 python -m pytest tests/test_gpu_fused_fullsize.py tests/test_reference_dropin.py -q -m gpu -s ) > $O/fullsize.log 2>&1; echo "fullsize rc=$?" >> $O/summary.txt
cat $O/summary.txt; cat $O/kbench.log; tail -5 $O/wave.log; tail -15 $O/parity_scan.log
