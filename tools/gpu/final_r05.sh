#!/bin/bash
# round 5, final tree: the -m gpu suite, the bench line, the kernel-trace of the step, the 2M-Gaussian bench
# (K8's FETCH / WRITE passes are not repeated: the kernel's sources are unchanged, profiles/pmc_k_render_bwd.json carries their hash)
cd ${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r05}
P=$PWD/gpurun_out/profiles; mkdir -p $P gpurun_out/tests; export TMPDIR=/tmp PYTHONUNBUFFERED=1
R=$PWD
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print(\"smoke ok\")" 2>&1 | tail -2
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -40 > gpurun_out/tests/pytest.log; tail -3 gpurun_out/tests/pytest.log
timeout 600 python bench.py --steps 20 --warmup 5 > $P/${TAG}_bench.json 2> $P/${TAG}_bench.err; echo "bench rc=$?"
B="python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-op-only --streams 1 --shard-views 0"
( cd /tmp && rm -rf /tmp/prof_kt && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_kt -o kt -- $B ) > $P/${TAG}_kt.log 2>&1; echo "kt rc=$?"
python - <<PY
import csv, glob
for f in glob.glob('/tmp/prof_kt/**/*kernel_stats.csv', recursive=True):
    rows = sorted(csv.DictReader(open(f)), key=lambda r: -float(r.get('TotalDurationNs', 0) or 0))
    with open('$P/${TAG}_kernel_stats.csv', 'w', newline='') as fo:
        w = csv.DictWriter(fo, fieldnames=rows[0].keys()); w.writeheader()
        for r in rows[:40]:
            r = dict(r); r['Name'] = r['Name'][:110]; w.writerow(r)
    for r in rows[:16]:
        print('KT %-60s calls %5s avg %9.1f us' % (r['Name'][:60], r['Calls'], float(r['AverageNs']) / 1e3))
PY
tail -c 2500 $P/${TAG}_bench.json
timeout 600 python bench.py --workload cfg5 --steps 10 --warmup 3 --no-cpu-baseline --no-op-only > $P/${TAG}_bench_cfg5_2M.json 2> $P/${TAG}_bench_cfg5.err; echo "bench cfg5 rc=$?"; tail -c 600 $P/${TAG}_bench_cfg5_2M.json
