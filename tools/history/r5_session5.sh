#!/bin/bash
# round 5: A/B of the first candidates' records through the scalar cache and of the position of the first bound
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r5s5
rm -rf $OUT && mkdir -p $OUT
cd $REPO
timeout 900 bash tools/variants_k.sh "slic_assign_dot<true, false|slic_assign_dot<false, false|slic_assign_dot<true, true" recs0 recs1 ph3 ph5 ph3s > $OUT/variants.txt 2>&1
grep -v "Segmentation" $OUT/variants.txt
