#!/bin/bash
# connectivity: small cases first (tight timeouts), then the SLIC / connectivity tests, then the stage time
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/conn_${1:-a}
rm -rf $OUT && mkdir -p $OUT
cd $REPO
timeout 60 python - > $OUT/small.log 2>&1 <<'PY'
import numpy as np, sys
sys.path.insert(0, '.')
from oracle import oracle
from pyimsegm_amd import superpixels as S
np.random.seed(0)
for shape, sp in (((100, 150, 3), 20), ((64, 64, 3), 8), ((40, 200, 3), 10), ((257, 130, 3), 16)):
    img = np.random.random(shape)
    a = S.segment_slic_img2d(img, sp, 0.2)
    b = oracle.segment_slic_img2d(img, sp, 0.2)
    print(shape, sp, 'equal', np.array_equal(a, b), int(a.max()) + 1, flush=True)
from pyimsegm_amd.utilities.synthetic import voronoi_image
for size, sp in ((256, 18), (600, 25), (1024, 30)):
    img = voronoi_image(size, size + 64, seed=3)
    a = S.segment_slic_img2d(img, sp, 0.2)
    b = oracle.segment_slic_img2d(img, sp, 0.2)
    print(size, sp, 'equal', np.array_equal(a, b), int(a.max()) + 1, 'diff px', int((a != b).sum()), flush=True)
PY
echo "small rc=$?"; cat $OUT/small.log | tail -12
timeout 200 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -8 $OUT/pytest.log
timeout 100 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --inflight 1 > $OUT/b1.json 2> $OUT/b1.err
python - $OUT <<'PY'
import json, sys, os
for n in ('b1',):
    try:
        d = json.loads(open(os.path.join(sys.argv[1], n + '.json')).read().strip().splitlines()[-1])
        print(n, d['value'], 'Mpx/s', d['ms_per_step'], 'ms', 'frac', d['roofline']['frac'], 'eq_ref', d.get('gpu_equals_reference_run'), 'lat', d.get('latency_ms'))
        print('   ', d['stage_ms_per_step'])
    except Exception as ex:
        print(n, 'ERR', ex)
PY
