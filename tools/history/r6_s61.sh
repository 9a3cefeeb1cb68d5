#!/bin/bash
# round 6, session 61: the label writes and the final gather four voxels per lane -- all GPU tests, config 5 kernel statistics
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
python -m pytest tests -m gpu -x -q > gpurun_out/pytest_s61.log 2>&1; grep -n "passed\|failed\|Error" gpurun_out/pytest_s61.log | tail -3
bash tools/c5_kstats.sh | grep "total kernel\|k_write_labels\|k_gather_i32"
