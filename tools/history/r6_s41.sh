#!/bin/bash
# round 6, session 41: the tiles' boxes by runs with OR-ed sets; parity + kernel statistics of config 5
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
python -m pytest tests/test_gpu_volume.py tests/test_gpu_fused_update.py -m gpu -x -q 2>&1 | tail -3
bash tools/c5_kstats.sh | head -12
