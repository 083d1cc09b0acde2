#!/bin/bash
# ordered kernel sequence of ONE step of the single-view bench (rocprofv3 --kernel-trace): tools/gpu/seq.sh [tag]
cd ${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-seq}; shift
P=$PWD/gpurun_out/kt; mkdir -p $P; export TMPDIR=/tmp PYTHONUNBUFFERED=1
R=$PWD
B="python $R/bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-op-only --streams 1 --shard-views 0"
( cd /tmp && rm -rf /tmp/prof_$TAG && env "$@" timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_$TAG -o kt -- $B ) > $P/${TAG}.log 2>&1; echo "kt rc=$?"
python - <<PY
import csv, glob
for f in glob.glob('/tmp/prof_$TAG/**/*kernel_trace.csv', recursive=True):
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
    # find the k_adam launches: a step ends with one
    idx = [i for i, r in enumerate(rows) if 'k_adam_finish' in r['Kernel_Name']]
    if len(idx) < 6: continue
    a, b = idx[-4] + 1, idx[-3] + 1
    t_prev = int(rows[a - 1]['End_Timestamp'])
    t0 = int(rows[a]['Start_Timestamp'])
    for r in rows[a:b]:
        s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        print('SEQ %8.1f us  +gap %6.1f  dur %7.1f  %s' % ((s - t0) / 1e3, (s - t_prev) / 1e3, (e - s) / 1e3, r['Kernel_Name'][:70]))
        t_prev = e
    print('SEQ step span %.1f us' % ((int(rows[b - 1]['End_Timestamp']) - int(rows[idx[-4]]['End_Timestamp'])) / 1e3))
PY
