#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
timeout 600 python -c "import __graft_entry__ as g; g.build(); g.smoke(); print('SMOKE OK')" 2>&1 | tail -2
timeout 600 python bench.py > /tmp/b.out 2>/tmp/b.err; echo "bench rc=$? lines=$(wc -l < /tmp/b.out)"; python -c "
import json; d=json.load(open('/tmp/b.out')); print(sorted(d.keys())); print(d['metric'], d['value'], d['unit'], d['n_gpus'], d['steps'], d['warmup'], d['ms_per_step'], d['higher_is_better'], d['scaling'], d['vs_baseline'], d['dtype'], d['data'])"
