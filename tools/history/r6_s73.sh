#!/bin/bash
# round 6, session 73: the scan of the arc / edge offsets with four entries per lane and turn -- all GPU tests, config 5 statistics
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
python -m pytest tests -m gpu -x -q > gpurun_out/pytest_s73.log 2>&1; grep -n "passed\|failed\|Error" gpurun_out/pytest_s73.log | tail -3
bash tools/c5_kstats.sh | grep "total kernel\|k_adj_scan\|k_scan_blocksums"
