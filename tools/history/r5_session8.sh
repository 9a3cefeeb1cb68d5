#!/bin/bash
# round 5: the volume pipeline with the pre-touched result array -- parity subset + config 5 at BASELINE size
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r5s8
rm -rf $OUT && mkdir -p $OUT
cd $REPO
timeout 400 python -m pytest tests/test_gpu_volume.py tests/test_gpu_zz_configs.py tests/test_gpu_fused.py -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
timeout 600 python bench.py --config 5 --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench_c5.json 2> $OUT/bench_c5.err; echo "bench rc=$?"
python - <<'P'
import json
d=json.loads(open('/root/repo/gpurun_out/r5s8/bench_c5.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d.get('host_model_fit_ms_per_step'), d.get('ms_per_step_excluding_fit'), d.get('gpu_slic_equals_scikit_image'), d['stage_ms_per_step'])
P
