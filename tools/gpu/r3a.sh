#!/bin/bash
# round 3, GPU call a: full -m gpu suite at ABI 13 + K8 ablations (kbench) + bench line + 2-rank gloo bench (scaling block)
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r3a; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 > $O/pytest.log
for v in "" noarith noatom lds20; do
  lib=""; [ -n "$v" ] && lib=$PWD/build/variants/libghr_$v.so
  for cfg in cfg3 cfg2; do
    GHR_LIB_PATH=$lib timeout 300 python tools/kbench.py $cfg 20 2>&1 | grep KBENCH | sed "s/^/[${v:-product}] /" >> $O/kbench.log
  done
done
GHR_LIB_PATH=$PWD/build/variants/libghr_prof.so GHR_PROF_NAMES=draw,pix,list,gwait,chunk,tail timeout 300 python tools/kbench.py cfg3 10 2>&1 | grep "PROF\|KBENCH" >> $O/kbench.log
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
GHR_BENCH_BACKEND=gloo GHR_BENCH_SHARE_GPU=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 5 --warmup 2 --no-op-only --no-cpu-baseline > $O/bench_gloo2.json 2> $O/bench_gloo2.err
tail -3 $O/pytest.log; cat $O/kbench.log; tail -c 600 $O/bench_gloo2.err
