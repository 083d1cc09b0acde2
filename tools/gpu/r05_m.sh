#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/q; export PYTHONUNBUFFERED=1
V=$PWD/build/variants
GHR_LIB_PATH=$V/libghr_pbwprof.so GHR_PROF_NAMES="first_trip,line_numbers+lines,barrier,compute+scalar_stores,barrier2,slab_out_issue,total,stores_acked" timeout 300 python tools/pbwd_prof.py > gpurun_out/q/pbw.log 2>&1; echo "pbw rc=$?"
GHR_LIB_PATH=$V/libghr_pk1prof.so GHR_PROF_NAMES="slab+raw_in+barrier,project_core,atomics_issue+scan,thread_stores+barrier,rec_stores_issue,atomics_collect+pos,total,stores_acked" timeout 300 python tools/pbwd_prof.py > gpurun_out/q/pk1.log 2>&1; echo "pk1 rc=$?"
grep -h "PHASES\|ms_per_step" gpurun_out/q/pbw.log gpurun_out/q/pk1.log | cut -c1-400
