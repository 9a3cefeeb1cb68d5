#!/bin/bash
# round 6, session 45: the raster-order update with the values of a quad loaded only where one of its labels is the lane's (variant
# lazyval) against loading them always; the cut after the relabelling's last changes
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fused.py -m gpu -x -q > gpurun_out/pytest_s45.log 2>&1; grep -n "passed\|failed" gpurun_out/pytest_s45.log
bash tools/c5_kstats.sh | grep "total kernel\|update_f32\|alpha_exp"
IMSEGM_HIP_LIBRARY=$REPO/pyimsegm_amd/build/variants/lazyval.so bash tools/c5_kstats.sh | grep "total kernel\|update_f32\|alpha_exp"
