#!/bin/bash
# PMC passes over a short bench run; prints mean counter values per dispatch of the k_slic_* kernels.
#   tools/pmc_assign.sh "SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES" ...
cd /tmp && export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
i=0
for set in "$@"; do
  i=$((i+1))
  out=$REPO/gpurun_out/pmcA$i
  rm -rf $out
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out -o p -- python $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --inflight 1 > $out.log 2>&1
  python - "$out/p_counter_collection.csv" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for row in csv.DictReader(open(sys.argv[1])):
    name = row['Kernel_Name']
    if 'k_slic' not in name:
        continue
    acc[name.split('(')[0]][row['Counter_Name']].append(float(row['Counter_Value']))
for k in sorted(acc):
    print(k[:60], ' '.join('%s=%.0f' % (c, sum(v) / len(v)) for c, v in sorted(acc[k].items())), 'n=%d' % len(next(iter(acc[k].values()))))
PY
done
