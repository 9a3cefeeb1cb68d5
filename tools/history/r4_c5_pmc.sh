#!/bin/bash
# SQ counters of the supervoxel kernels on a 64 x 1024 x 1024 volume (1/16 of config 5: the same bricks and windows)
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/c5pmc; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for set in ${PMC_SETS:-"SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAVES SQ_BUSY_CYCLES"} ; do
  rm -rf $OUT/p
  timeout 600 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/p -o p -- python $REPO/bench.py --config 5 --volume 64,1024,1024 --steps 1 --warmup 1 --no-cpu-baseline > $OUT/pmc.log 2>&1
  python - $OUT/p <<'PY'
import csv, sys, collections, glob
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True):
    for row in csv.DictReader(open(f)):
        n = row['Kernel_Name']
        if 'k_vol_assign' not in n and 'k_vol_update' not in n: continue
        acc[n.split('(')[0]][row['Counter_Name']].append(float(row['Counter_Value']))
for k in sorted(acc):
    print(k[:50], ' '.join('%s=%.3g' % (c, sum(v) / len(v)) for c, v in sorted(acc[k].items())), 'n=%d' % len(next(iter(acc[k].values()))))
PY
done
