#!/bin/bash
# round 6, session 55: volume tests + kernel statistics of config 5 on the library as committed
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
python -m pytest tests/test_gpu_volume.py tests/test_gpu_fused_update.py tests/test_gpu_zz_configs.py -m gpu -x -q 2>&1 | grep "passed\|failed"
bash tools/c5_kstats.sh | head -24 | cut -c1-150
