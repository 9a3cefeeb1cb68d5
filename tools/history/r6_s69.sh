#!/bin/bash
# round 6, session 69: the texture kernels' tiles requested in one go -- parity tests, config 3 line and kernel statistics
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py tests/test_gpu_texture_size.py tests/test_gpu_resident_table.py tests/test_gpu_zz_configs.py tests/test_gpu_zz_reference.py -m gpu -x -q > gpurun_out/pytest_s69.log 2>&1; grep -n "passed\|failed\|Error" gpurun_out/pytest_s69.log | tail -3
cd /tmp && export TMPDIR=/tmp
OUT=$REPO/gpurun_out/c3ks; rm -rf $OUT; mkdir -p $OUT
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/kt -o bench -- python $REPO/bench.py --config 3 --steps 3 --warmup 1 --inflight 1 --no-cpu-baseline --no-other-configs > $OUT/bench.json 2> $OUT/err.log
DB=$(find $OUT/kt -name "*.db" | head -1); python $REPO/tools/prof_summary.py $DB > $OUT/kernel_stats.txt; rm -rf $OUT/kt
head -9 $OUT/kernel_stats.txt | cut -c1-150
python - <<'P'
import json
d=json.loads(open('/root/repo/gpurun_out/c3ks/bench.json').read().strip().splitlines()[-1])
print('config 3', d['value'], d['ms_per_step'], d['roofline']['frac'])
P
