#!/bin/bash
# kernel timelines of config 4 (12 in flight) and config 2 (3 in flight): how busy the device is (tools/trace_overlap.py)
cd /tmp && export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
out=$REPO/gpurun_out/overlap
mkdir -p $out
for c in ${CONFIGS:-4 2}; do
  rm -rf $out/kt$c
  rocprofv3 --kernel-trace --output-format csv -d $out/kt$c -o p -- python $REPO/bench.py --config $c --steps ${STEPS:-30} --warmup 2 --no-cpu-baseline --no-other-configs ${EXTRA} > $out/bench$c.log 2>&1
  f=$(find $out/kt$c -name "*kernel_trace.csv" | head -1)
  echo "== config $c"; python $REPO/tools/trace_overlap.py $f 0.3 | tee $out/overlap_cfg$c.txt
  rm -rf $out/kt$c
done
