#!/bin/bash
# round 6: PMC counters of EVERY kernel of the volume pipeline (config 5), one rocprofv3 pass per counter set (FETCH_SIZE and
# WRITE_SIZE cannot share a pass), no tracing domain beside --kernel-trace.  Volume: $1 (default 64,2048,2048 = a quarter of config 5:
# same bricks, same windows, 4.3 GB of planes >> the Infinity Cache).  Summary -> gpurun_out/c5pmc/summary.{txt,json}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
VOL=${1:-64,2048,2048}
OUT=$REPO/gpurun_out/c5pmc; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_WAVE_CYCLES" "SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM" "TCC_HIT_sum TCC_MISS_sum" ; do
  i=$((i+1))
  rm -rf $OUT/p
  timeout 900 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/p -o p -- python $REPO/bench.py --config 5 --volume $VOL --steps 1 --warmup 0 --inflight 1 --no-cpu-baseline > $OUT/pmc$i.log 2>&1
  echo "set $i ($set) rc=$?"
  f=$(find $OUT/p -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && cp $f $OUT/counters_$i.csv
done
rm -rf $OUT/p
python $REPO/tools/pmc_table.py $OUT $VOL
