#!/bin/bash
# round 6, session 40: 16 x 16 tiles (boxes label by label) against 64 x 4 strips in the float32 3-D assignment -- parity tests with both, kernel statistics
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
python -m pytest tests/test_gpu_volume.py tests/test_gpu_fused_update.py -m gpu -x -q 2>&1 | tail -3
IMSEGM_HIP_LIBRARY=$REPO/pyimsegm_amd/build/variants/strip64.so python -m pytest tests/test_gpu_volume.py -m gpu -x -q -k "slic" 2>&1 | tail -3
bash tools/c5_kstats.sh | head -12
cp gpurun_out/c5ks/kernel_stats.txt gpurun_out/c5ks/kernel_stats_tile16.txt
export IMSEGM_HIP_LIBRARY=$REPO/pyimsegm_amd/build/variants/strip64.so
bash tools/c5_kstats.sh | head -8
cp gpurun_out/c5ks/kernel_stats.txt gpurun_out/c5ks/kernel_stats_strip64.txt
