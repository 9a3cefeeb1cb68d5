#!/bin/bash
# round 6, session 42: the grid-wide cut's relabelling by the frontier -- parity tests, kernel statistics of config 5
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fused.py tests/test_gpu_zz_configs.py -m gpu -x -q 2>&1 | tail -3
bash tools/c5_kstats.sh | head -12
IMSEGM_GC_DEBUG=1 python bench.py --config 5 --steps 1 --warmup 1 --inflight 1 --no-cpu-baseline 2>&1 | grep "alpha expansion" | tail -1
