#!/bin/bash
# Round 3 profile set: the round profile of tools/profile_round.sh (default bench line, rocprofv3 kernel statistics, PMC passes) plus
# the persistent-sweep variants, the other configs on their own, and two self-spawned ranks sharing the one GPU of the box.
TAG=${1:-r03}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
bash $REPO/tools/profile_round.sh $TAG > /dev/null 2>&1
OUT=$REPO/gpurun_out/prof_$TAG
cd $REPO
for g in 9 3 1; do
  IMSEGM_SLIC_PERSISTENT=1 IMSEGM_SWEEPS_PER_LAUNCH=$g timeout 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-other-configs --inflight 1 > $OUT/bench_persistent_g${g}_inflight1.json 2>> $OUT/persistent.err
done
IMSEGM_SLIC_PERSISTENT=1 timeout 200 python bench.py --steps 40 --warmup 3 --no-cpu-baseline --no-other-configs > $OUT/bench_persistent_g9.json 2>> $OUT/persistent.err
timeout 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-other-configs --inflight 1 > $OUT/bench_per_sweep_inflight1.json 2>> $OUT/persistent.err
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
for f in $OUT/bench*.json; do echo "$(basename $f): $(python tools/bl.py < $f 2>/dev/null || tail -c 300 $f)"; done
tail -c 400 $OUT/bench_gpus2.err
