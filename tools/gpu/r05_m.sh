#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/q2; export PYTHONUNBUFFERED=1
V=$PWD/build/variants
GHR_LIB_PATH=$V/libghr_pbwprof.so GHR_PROF_NAMES="first_trip,line_numbers+lines,geometry(slab_in_flight),slab_wait_rest,sh+scalar_stores,slab_out_issue,total,stores_acked" timeout 300 python tools/pbwd_prof.py > gpurun_out/q2/pbw.log 2>&1; echo "pbw rc=$?"
grep -h "PHASES" gpurun_out/q2/pbw.log | cut -c1-400
