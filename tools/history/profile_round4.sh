#!/bin/bash
# Round-4 / round-5 profile set (TAG = first argument): the driver's command, rocprofv3 kernel statistics of configs 2 / 3 / 4 (8 images per launch chain), the
# HBM-traffic PMC passes of the assignment kernel for one 2048^2 image and for a batch of eight 647 x 1024 images, two ranks on
# the one GPU of the box.  Run on the GPU box (gpurun); summaries land in gpurun_out/prof_<tag>/ and are copied into profiles/.
TAG=${1:-r04}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_$TAG
rm -rf $OUT && mkdir -p $OUT
cd $REPO
python bench.py > $OUT/bench.json 2> $OUT/bench.err                  # the default command: what the driver runs
python bench.py --gpus 2 --config 4 --steps 12 --no-cpu-baseline > $OUT/bench_cfg4_gpus2_one_device.json 2> $OUT/bench_gpus2.err
cd /tmp && export TMPDIR=/tmp
kstats() {   # name, bench arguments...
  name=$1; shift
  rm -rf $OUT/kt
  rocprofv3 --kernel-trace --stats -d $OUT/kt -o bench -- python $REPO/bench.py "$@" --no-cpu-baseline > $OUT/bench_rocprof_$name.json 2> $OUT/kt_$name.err
  DB=$(find $OUT/kt -name "*.db" | head -1)
  python $REPO/tools/prof_summary.py $DB > $OUT/kernel_stats_$name.txt
  rm -rf $OUT/kt
}
kstats cfg2 --steps 10 --warmup 2 --no-other-configs
kstats cfg2_inflight1 --steps 10 --warmup 2 --no-other-configs --inflight 1
kstats cfg4_batch --config 4 --steps 12 --warmup 2
kstats cfg4_batch_inflight1 --config 4 --steps 12 --warmup 2 --inflight 1
kstats cfg3 --config 3 --steps 3 --warmup 1
pmc() {      # name, counters, bench arguments...
  name=$1; counters=$2; shift; shift
  rm -rf $OUT/pmc_$name
  rocprofv3 --pmc $counters --kernel-trace --output-format csv -d $OUT/pmc_$name -o p -- python $REPO/bench.py "$@" --no-cpu-baseline --inflight 1 > $OUT/pmc_$name.log 2>&1
}
pmc c2_FETCH FETCH_SIZE --steps 2 --warmup 1 --no-other-configs
pmc c2_WRITE WRITE_SIZE --steps 2 --warmup 1 --no-other-configs
pmc c2_SQ "SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVES" --steps 2 --warmup 1 --no-other-configs
pmc c4_FETCH FETCH_SIZE --config 4 --steps 3 --warmup 1
pmc c4_WRITE WRITE_SIZE --config 4 --steps 3 --warmup 1
python - $OUT $TAG <<'PY'
import csv, sys, collections, json, os
out = sys.argv[1]
tag_name = sys.argv[2] if len(sys.argv) > 2 else 'r04'
def table(path, zmin=None):
    # (zmin: keep the launches of a whole batch only -- per kernel the dispatches with the largest grid)
    rows = list(csv.DictReader(open(path)))
    def grid(row):
        if 'Grid_Size' in row and row['Grid_Size']:
            return int(row['Grid_Size'])
        return int(row.get('Grid_Size_X', 1) or 1) * int(row.get('Grid_Size_Y', 1) or 1) * int(row.get('Grid_Size_Z', 1) or 1)
    largest = collections.defaultdict(int)
    for row in rows:
        name = row['Kernel_Name'].split('(')[0]
        largest[name] = max(largest[name], grid(row))
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for row in rows:
        name = row['Kernel_Name'].split('(')[0]
        if zmin is not None and grid(row) < largest[name]:
            continue
        acc[name][row['Counter_Name']].append(float(row['Counter_Value']))
    return acc
lines, res = [], {}
for tag, zmin, what in (('c2', None, 'bench.py --steps 2 --warmup 1 --inflight 1, one 2048x2048 image per launch'),
                        ('c4', 8, 'bench.py --config 4 --steps 3 --warmup 1 --inflight 1, launches of 8 images of 647x1024 only (grid z = 8)')):
    for c in ('FETCH', 'WRITE', 'SQ'):
        p = os.path.join(out, 'pmc_%s_%s' % (tag, c), 'p_counter_collection.csv')
        if not os.path.exists(p):
            continue
        try:
            acc = table(p, zmin)
        except Exception as ex:
            lines.append('## %s %s: %r' % (tag, c, ex))
            continue
        lines.append('## rocprofv3 --pmc pass %s (%s; mean per dispatch)' % (c, what))
        for k in sorted(acc):
            lines.append('%-64s %s n=%d' % (k[:64], ' '.join('%s=%.0f' % (n, sum(v) / len(v)) for n, v in sorted(acc[k].items())),
                                            len(next(iter(acc[k].values())))))
            res.setdefault(tag, {}).setdefault(k, {}).update({n: sum(v) / len(v) for n, v in acc[k].items()})
        lines.append('')
open(os.path.join(out, 'pmc_counters.txt'), 'w').write('\n'.join(lines))
for tag, fname, workload in (('c2', 'pmc_traffic.json', 'bench.py default (2048x2048 RGB, K=2025), one image per launch'),
                             ('c4', 'pmc_traffic_cfg4_batch.json', 'bench.py --config 4: eight 647x1024 images per launch (grid z = 8)')):
    name = [k for k in res.get(tag, {}) if 'k_slic_assign_dot<true, false' in k]
    if name and 'FETCH_SIZE' in res[tag][name[0]] and 'WRITE_SIZE' in res[tag][name[0]]:
        f, w = res[tag][name[0]]['FETCH_SIZE'], res[tag][name[0]]['WRITE_SIZE']
        json.dump({'kernel': 'k_slic_assign_dot<true, false>', 'workload': workload, 'sweeps_per_launch': 1,
                   'fetch_size_kb': f, 'write_size_kb': w, 'hbm_bytes_per_launch': int((2 * f + w) * 1024),
                   'how': 'rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (profiles/rocprof_%s_pmc_counters.txt); ' % tag_name +
                          'FETCH_SIZE doubled per the gfx950 wide-coalesced-read correction of MI355X_MICROARCH.md; (2*fetch + write) KB'},
                  open(os.path.join(out, fname), 'w'), indent=2)
PY
rm -rf $OUT/pmc_c2_FETCH $OUT/pmc_c2_WRITE $OUT/pmc_c2_SQ $OUT/pmc_c4_FETCH $OUT/pmc_c4_WRITE
tail -c 600 $OUT/bench.err; head -c 700 $OUT/bench.json; echo; head -14 $OUT/kernel_stats_cfg2_inflight1.txt | cut -c1-140; cat $OUT/pmc_traffic*.json
