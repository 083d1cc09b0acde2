#!/bin/bash
# round 6, first trip: camera-gradient tests (ABI 17), the whole -m gpu suite, the bench line of the untouched step
cd ${GRAFT_REPO_ROOT:-/root/repo}
P=$PWD/gpurun_out/r06a; mkdir -p $P; export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 600 python -m pytest tests/test_camera_grads.py -m gpu -x -q -s 2>&1 | tail -40 > $P/cam_tests.log; tail -15 $P/cam_tests.log
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -40 > $P/pytest.log; tail -5 $P/pytest.log
timeout 600 python bench.py --steps 20 --warmup 5 > $P/bench.json 2> $P/bench.err; echo "bench rc=$?"; tail -c 1500 $P/bench.json
