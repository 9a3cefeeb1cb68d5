#!/bin/bash
# Round profile: bench JSON, rocprofv3 kernel stats and the HBM-traffic PMC passes of the same command.
# Run on the GPU box (gpurun); copies the summaries into profiles/ under the given tag.
TAG=${1:-r02a}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
OUT=$REPO/gpurun_out/prof_$TAG
rm -rf $OUT && mkdir -p $OUT
python $REPO/bench.py > $OUT/bench.json 2> $OUT/bench.err          # the default command: what the driver runs
# the default command (three images in flight: kernels of different streams overlap) ...
rocprofv3 --kernel-trace --stats -d $OUT/kt -o bench -- python $REPO/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-other-configs > $OUT/bench_rocprof_run.json 2> $OUT/kt.err
DB=$(find $OUT/kt -name "*.db" | head -1)
python $REPO/tools/prof_summary.py $DB > $OUT/kernel_stats.txt
# ... and one image at a time: these per-kernel durations are the ones bench.py's roofline pass measures
rocprofv3 --kernel-trace --stats -d $OUT/kt1 -o bench -- python $REPO/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-other-configs --inflight 1 > $OUT/bench_rocprof_run_inflight1.json 2> $OUT/kt1.err
DB=$(find $OUT/kt1 -name "*.db" | head -1)
python $REPO/tools/prof_summary.py $DB > $OUT/kernel_stats_inflight1.txt
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_$c -o p -- python $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs --inflight 1 > $OUT/pmc_$c.log 2>&1
done
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVES --kernel-trace --output-format csv -d $OUT/pmc_SQ -o p -- python $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs --inflight 1 > $OUT/pmc_SQ.log 2>&1
python - $OUT <<'PY'
import csv, sys, collections, json, os
out = sys.argv[1]
def table(path):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for row in csv.DictReader(open(path)):
        acc[row['Kernel_Name'].split('(')[0]][row['Counter_Name']].append(float(row['Counter_Value']))
    return acc
lines = []
res = {}
for c in ('FETCH_SIZE', 'WRITE_SIZE', 'SQ'):
    p = os.path.join(out, 'pmc_' + c, 'p_counter_collection.csv')
    if not os.path.exists(p):
        continue
    acc = table(p)
    lines.append('## rocprofv3 --pmc pass %s (mean per dispatch; bench.py --steps 2 --warmup 1 --inflight 1, 2048x2048)' % c)
    for k in sorted(acc):
        lines.append('%-64s %s n=%d' % (k[:64], ' '.join('%s=%.0f' % (n, sum(v) / len(v)) for n, v in sorted(acc[k].items())),
                                        len(next(iter(acc[k].values())))))
        res.setdefault(k, {}).update({n: sum(v) / len(v) for n, v in acc[k].items()})
    lines.append('')
open(os.path.join(out, 'pmc_counters.txt'), 'w').write('\n'.join(lines))
name = [k for k in res if 'k_slic_assign_dot<true, false' in k]
if name and 'FETCH_SIZE' in res[name[0]] and 'WRITE_SIZE' in res[name[0]]:
    f, w = res[name[0]]['FETCH_SIZE'], res[name[0]]['WRITE_SIZE']
    json.dump({'kernel': 'k_slic_assign_dot<true, false>', 'workload': 'bench.py default (2048x2048 RGB, K=2025)',
               'fetch_size_kb': f, 'write_size_kb': w, 'hbm_bytes_per_launch': int((2 * f + w) * 1024),
               'how': 'rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (profiles/rocprof_*_pmc_counters.txt); '
                      'FETCH_SIZE doubled per the gfx950 wide-coalesced-read correction of MI355X_MICROARCH.md; (2*fetch + write) KB'},
              open(os.path.join(out, 'pmc_traffic.json'), 'w'), indent=2)
PY
cat $OUT/bench.json
head -12 $OUT/kernel_stats_inflight1.txt
grep "k_slic_assign_dot" $OUT/pmc_counters.txt
