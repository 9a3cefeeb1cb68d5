#!/bin/bash
# round 5: stability of the side-by-side fit on the GPU box -- config 5 at BASELINE size three times (nine fits), + the small volume tests
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r5s9
rm -rf $OUT && mkdir -p $OUT
cd $REPO
timeout 300 python -m pytest tests/test_gpu_volume.py tests/test_gpu_zz_configs.py -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $OUT/pytest.log
for i in 1; do
  timeout 400 python bench.py --config 5 --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench_c5_$i.json 2> $OUT/bench_c5_$i.err; echo "bench $i rc=$?"
  python - $i <<'P'
import json, sys
d=json.loads(open('/root/repo/gpurun_out/r5s9/bench_c5_%s.json' % sys.argv[1]).read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d.get('host_model_fit_ms_per_step'), d.get('ms_per_step_excluding_fit'), d.get('gpu_slic_equals_scikit_image'))
P
  grep -c "NUM_THREADS" $OUT/bench_c5_$i.err
done
