#!/bin/bash
# round 6 rocprofv3 passes on the GPU box -> gpurun_out/profiles/ (summaries are copied to profiles/ and committed):
#  (1) the bench line (bench.py default), (2) --kernel-trace --stats of the step on one stream,
#  (3) FETCH_SIZE / WRITE_SIZE of k_render_bwd_cells on cfg3, cfg2 and cfg5 (+ the hash of the kernel's sources: bench.py's traffic_stale) (separate --pmc passes, kernel micro-bench),
#  (4) SQ counters of K7 / K8 (two passes of eight counters).
cd ${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r06}
P=$PWD/gpurun_out/profiles; mkdir -p $P; export TMPDIR=/tmp PYTHONUNBUFFERED=1
R=$PWD
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > /dev/null 2>&1; timeout 600 python bench.py --steps 20 --warmup 5 > $P/${TAG}_bench.json 2> $P/${TAG}_bench.err; echo "bench rc=$?"
B="python $R/bench.py --steps 16 --warmup 16 --no-cpu-baseline --no-op-only --no-2m --no-camera-block --no-strand-block --streams 1 --shard-views 0"
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
for cfg in cfg3 cfg2 cfg5; do for ctr in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && rm -rf /tmp/pmc_${cfg}_$ctr && timeout 150 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d /tmp/pmc_${cfg}_$ctr -o p -- python $R/tools/kbench.py $cfg 5 ) > $P/${TAG}_pmc_${cfg}_$ctr.log 2>&1; echo "pmc $cfg $ctr rc=$?"
done; done
python - <<PY
import csv, glob, json, sys
sys.path.insert(0, '.')
from tools.k8_source_hash import k8_source_hash, FILES
out = {"kernel": "k_render_bwd_cells", "k8_source_sha256": k8_source_hash(), "k8_source_files": list(FILES), "correction": "(2*FETCH_SIZE + WRITE_SIZE) * 1024: counters in KiB; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports half of wide coalesced reads); other widths uncalibrated",
       "source": "profiles/${TAG}: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over tools/kbench.py"}
for cfg in ('cfg3', 'cfg2', 'cfg5'):
    v = {}
    for ctr in ('FETCH_SIZE', 'WRITE_SIZE'):
        vals = []
        for f in glob.glob('/tmp/pmc_%s_%s/**/*counter_collection.csv' % (cfg, ctr), recursive=True):
            for r in csv.DictReader(open(f)):
                if 'k_render_bwd' in r['Kernel_Name'] and r['Counter_Name'] == ctr:
                    vals.append(float(r['Counter_Value']))
        if vals:
            v[ctr + '_KiB'] = sum(vals) / len(vals); v[ctr + '_launches'] = len(vals)
    if 'FETCH_SIZE_KiB' in v and 'WRITE_SIZE_KiB' in v:
        v['hbm_bytes_per_launch'] = (2 * v['FETCH_SIZE_KiB'] + v['WRITE_SIZE_KiB']) * 1024
        v['source'] = out['source']
    out[cfg] = v
json.dump(out, open('$P/pmc_k_render_bwd.json', 'w'), indent=1)
print('PMC', json.dumps(out))
PY
rm -f $P/${TAG}_sq_counters.txt
bash tools/gpu/sqpmc.sh $P/${TAG}_sq_counters.txt cells cfg3
cat $P/${TAG}_sq_counters.txt; tail -c 1500 $P/${TAG}_bench.json

