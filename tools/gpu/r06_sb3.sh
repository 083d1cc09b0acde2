#!/bin/bash
# round 6: strand-stage iteration, HEAD (848c0c9: cat / copy glue, in-place late groups) against the working tree, same box
cd ${GRAFT_REPO_ROOT:-/root/repo}
R=$PWD; O=$PWD/gpurun_out/r06sb3; mkdir -p $O; export TMPDIR=/tmp PYTHONUNBUFFERED=1
rm -f $O/strand_ab.log
for rep in 1 2 3; do
python build/head_tree/tools/strandstep.py 40 2>&1 | grep STRAND | sed 's/^/[848c0c9] /' | tee -a $O/strand_ab.log
python tools/strandstep.py 40 2>&1 | grep STRAND | sed 's/^/[compact outputs, out-of-place late groups, no property evaluation] /' | tee -a $O/strand_ab.log
done
