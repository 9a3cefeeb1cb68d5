#!/bin/bash
# round 5: A/B of the DPP preset in the accumulation, the exact constant division and the tile height of the pre-processing kernel
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r5s6
rm -rf $OUT && mkdir -p $OUT
cd $REPO
timeout 300 python -m pytest tests/test_gpu_golden.py tests/test_gpu_zz_skimage.py tests/test_gpu_sweeps.py tests/test_gpu_batch.py tests/test_gpu_zz_configs.py -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
timeout 900 bash tools/variants_k.sh "slic_assign_dot<true, false|k_pre_fused" nodiv cur preset pf32 pf48 > $OUT/variants.txt 2>&1
grep -v "Segmentation" $OUT/variants.txt
