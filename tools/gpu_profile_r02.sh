#!/bin/bash
# round 2 rocprofv3 passes on the GPU box -> gpurun_out/profiles/ (summaries are copied to profiles/ and committed):
#  (1) --kernel-trace --stats of the bench step on one stream (per-kernel averages),
#  (2) FETCH_SIZE / WRITE_SIZE of k_render_bwd on cfg3 and cfg5 (separate --pmc passes, kernel micro-bench),
#  (3) memory-path counters of the projection kernels, two counters per pass.
mkdir -p gpurun_out/profiles; export TMPDIR=/tmp PYTHONUNBUFFERED=1
TAG=${1:-r02a}
R=$GRAFT_REPO_ROOT
P=gpurun_out/profiles
B="python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-op-only --streams 1 --shard-views 0"
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_kt -o kt -- $B ) > $P/${TAG}_kt.log 2>&1; echo "kt rc=$?"
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
for cfg in cfg3 cfg5; do for ctr in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && timeout 300 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d /tmp/pmc_${cfg}_$ctr -o p -- python $R/tools/kbench.py $cfg 5 ) > $P/${TAG}_pmc_${cfg}_$ctr.log 2>&1; echo "pmc $cfg $ctr rc=$?"
done; done
python - <<PY
import csv, glob, json, collections
out = {"kernel": "k_render_bwd", "correction": "(2*FETCH_SIZE + WRITE_SIZE) * 1024: counters in KiB; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports half of wide coalesced reads); other widths uncalibrated",
       "source": "profiles/${TAG}: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over tools/kbench.py"}
for cfg in ('cfg3', 'cfg5'):
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
i=0
for set in "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"; do
  i=$((i+1))
  ( cd /tmp && timeout 240 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pmem$i -o p -- $B ) > $P/${TAG}_pmem$i.log 2>&1; echo "pmem$i rc=$?"
  python - <<PY >> $P/${TAG}_mempath_counters.txt
import csv,glob,collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for fn in glob.glob('/tmp/pmem$i/**/*counter_collection.csv',recursive=True):
    for r in csv.DictReader(open(fn)):
        k=r['Kernel_Name'].split('(')[0].replace('ghr::','')
        if k.startswith('k_'): acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k in sorted(acc):
    print('PMC', k, ' '.join('%s=%.4g' % (c, sum(v)/len(v)) for c,v in acc[k].items()))
PY
done
cat $P/${TAG}_mempath_counters.txt
