#!/bin/bash
# round 5, third GPU session: SLIC parity subset, A/B (round-4 loop / round-5 loop with row masks / six waves per SIMD), SQ counters
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r5s3
rm -rf $OUT && mkdir -p $OUT
cd $REPO
timeout 300 python -m pytest tests/test_gpu_golden.py tests/test_gpu_zz_skimage.py tests/test_gpu_sweeps.py tests/test_gpu_batch.py tests/test_gpu_zz_configs.py tests/test_gpu_api.py -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
timeout 600 bash tools/variants_k.sh "slic_assign_dot<true, false|slic_assign_dot<false, false|k_slic_bin" base new3 occ6 > $OUT/variants.txt 2>&1
grep -v "Segmentation" $OUT/variants.txt
timeout 300 bash tools/pmc_assign.sh "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY" > $OUT/pmc.txt 2>&1
grep -v Segm $OUT/pmc.txt
