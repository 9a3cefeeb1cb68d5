#!/bin/bash
# round 5, second GPU session: A/B of the assignment kernel (round-4 loop / round-5 loop / + scalar-base addressing) and SQ counters
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r5s2
rm -rf $OUT && mkdir -p $OUT
cd $REPO
timeout 200 python -m pytest tests/test_gpu_golden.py tests/test_gpu_zz_skimage.py tests/test_gpu_sweeps.py tests/test_gpu_batch.py -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
timeout 600 bash tools/variants_k.sh "slic_assign_dot<true, false" base nosaddr new2 > $OUT/variants.txt 2>&1
grep -v "Segmentation" $OUT/variants.txt
timeout 400 bash tools/pmc_assign.sh "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INST_CYCLES_SALU SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM" > $OUT/pmc.txt 2>&1
cat $OUT/pmc.txt
