#!/bin/bash
# round 6, session 65: four quads per round in the update as the library's default -- all GPU tests, config 5 kernel statistics
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
python -m pytest tests -m gpu -x -q > gpurun_out/pytest_s65.log 2>&1; grep -n "passed\|failed\|Error" gpurun_out/pytest_s65.log | tail -3
bash tools/c5_kstats.sh | head -12 | cut -c1-150
cp gpurun_out/c5ks/kernel_stats.txt gpurun_out/c5_final_kstats.txt; cp gpurun_out/c5ks/bench.json gpurun_out/c5_final_bench.json
