#!/bin/bash
# round 6, session 50: the 3-D adjacency walking eight slices per workgroup -- parity, kernel statistics and counters of config 5
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
python -m pytest tests/test_gpu_volume.py tests/test_gpu_fused.py tests/test_gpu_zz_configs.py -m gpu -x -q > gpurun_out/pytest_s50.log 2>&1; grep -n "passed\|failed\|Error" gpurun_out/pytest_s50.log | tail -3
bash tools/c5_kstats.sh | grep "total kernel\|adjacency"
bash tools/c5_pmc.sh > gpurun_out/c5pmc_run.log 2>&1; grep -c . gpurun_out/c5pmc/summary.txt
