#!/bin/bash
# functional check of the N = 2 bench path on one GPU (two ranks share cuda:0, gloo): never a performance number
O=gpurun_out/r3g; mkdir -p $O; export TMPDIR=/tmp PYTHONUNBUFFERED=1
( GHR_BENCH_BACKEND=gloo GHR_BENCH_SHARE_GPU=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 5 --warmup 2 --no-cpu-baseline --no-op-only ) > $O/bench2.json 2> $O/bench2.err; echo "rc=$?"; tail -c 1500 $O/bench2.json; tail -5 $O/bench2.err
