#!/bin/bash
# round 5: kernel breakdown of the rasterizer op alone (cfg2, cfg3); the default-build rotated live tests with their counts printed
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp PYTHONUNBUFFERED=1
bash tools/gpu/opstats.sh cfg2
bash tools/gpu/opstats.sh cfg3
timeout 600 python -m pytest tests/test_gpu_reference_live.py -m gpu -q -s -k "default_build" 2>&1 | grep -E "live, default|passed|failed|Error" | cut -c1-400
