#!/bin/bash
# round 4: kernel statistics + SQ counters of the Leung-Malik kernels alone (one image, nothing else in flight)
TAG=${1:-c3pmc}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/$TAG; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/kt -o t -- python $REPO/tools/bench_texture.py > $OUT/run.log 2>&1
python $REPO/tools/prof_summary.py $(find $OUT/kt -name "*.db" | head -1) | head -14 | cut -c1-140
rm -rf $OUT/kt
for set in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAVES SQ_BUSY_CYCLES" "SQ_INSTS_SMEM SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SMEM SQ_WAIT_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_INSTS_SALU SQ_INST_CYCLES_SALU"; do
  rm -rf $OUT/p
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/p -o p -- python $REPO/tools/bench_texture.py > $OUT/pmc.log 2>&1
  python - $OUT/p <<'PY'
import csv, sys, collections, glob
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True):
    for row in csv.DictReader(open(f)):
        n = row['Kernel_Name']
        if 'battery' not in n and 'color_stats' not in n: continue
        acc[n.split('(')[0]][row['Counter_Name']].append(float(row['Counter_Value']))
for k in sorted(acc):
    print(k[:50], ' '.join('%s=%.3g' % (c, sum(v) / len(v)) for c, v in sorted(acc[k].items())), 'n=%d' % len(next(iter(acc[k].values()))))
PY
done
tail -3 $OUT/run.log
