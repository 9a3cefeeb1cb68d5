#!/bin/bash
# kernel-trace of a short bench run; prints mean duration of the k_slic_* kernels (or all with ALL=1)
cd /tmp && export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
out=$REPO/gpurun_out/ktrace
rm -rf $out
rocprofv3 --kernel-trace --output-format csv -d $out -o p -- python $REPO/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-other-configs --inflight ${INFLIGHT:-1} > $out.log 2>&1
python - "$out/p_kernel_trace.csv" <<'PY'
import csv, sys, collections, os
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    n = r['Kernel_Name']
    if 'k_slic' in n or os.environ.get('ALL'):
        acc[n.split('(')[0]].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1000)
for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
    print('%-60s n=%4d avg=%8.1f min=%8.1f max=%8.1f total=%9.1f' % (k[:60], len(v), sum(v) / len(v), min(v), max(v), sum(v)))
PY
