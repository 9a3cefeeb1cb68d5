#!/bin/bash
# Second profile set of round 3 (after the launch merges, the alpha-expansion rewrite, eight hardware queues / four images in flight,
# the recycled volume session): the round profile of tools/profile_round.sh, the other configs on their own, two self-spawned
# ranks on the one GPU of the box, kernel statistics of configs 3 and 4, and the device-occupancy timelines (tools/trace_overlap.py).
TAG=${1:-r03b}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
bash $REPO/tools/profile_round.sh $TAG > /dev/null 2>&1
OUT=$REPO/gpurun_out/prof_$TAG
cd $REPO
for c in 3 4; do timeout 300 python bench.py --config $c > $OUT/bench_cfg$c.json 2> $OUT/bench_cfg$c.err; done
timeout 900 python bench.py --config 5 > $OUT/bench_cfg5.json 2> $OUT/bench_cfg5.err
timeout 300 python bench.py --gpus 2 --steps 10 --warmup 2 > $OUT/bench_gpus2_one_device.json 2> $OUT/bench_gpus2.err
cd /tmp && export TMPDIR=/tmp
for c in 3 4; do
  rocprofv3 --kernel-trace --stats -d $OUT/kt_cfg$c -o bench -- python $REPO/bench.py --config $c --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> $OUT/kt_cfg$c.err
  DB=$(find $OUT/kt_cfg$c -name "*.db" | head -1)
  python $REPO/tools/prof_summary.py $DB > $OUT/kernel_stats_cfg$c.txt
  rm -rf $OUT/kt_cfg$c
done
rm -rf $OUT/kt $OUT/kt1 $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE $OUT/pmc_SQ
cd $REPO
STEPS=30 bash tools/r3_overlap.sh > $OUT/overlap.txt 2>&1
for f in $OUT/bench*.json; do echo "$(basename $f): $(tail -1 $f | python tools/bl.py 2>/dev/null || tail -c 300 $f)"; done
tail -c 300 $OUT/bench_gpus2.err
