#!/bin/bash
# A/B: pre-processing as ONE fused kernel (default) against TWO kernels (IMSEGM_PRE_2PASS): parity of the Lab planes and kernel times
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
IMSEGM_PRE_2PASS=1 timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | grep -E "passed|failed"
for v in fused two fused two; do
  if [ $v = two ]; then export IMSEGM_PRE_2PASS=1; else unset IMSEGM_PRE_2PASS; fi
  bash tools/kstats.sh ab_$v 200 2>/dev/null | grep -E "k_pre_fused|k_pre_lab|k_blur_yx|total kernel" | cut -c1-140
  python -c "
import json; d=json.loads(open('gpurun_out/ks_ab_$v/b1.json').read().strip().splitlines()[-1]); print('$v', d['value'], d['stage_ms_per_step']['slic_preprocess'], d['latency_ms'])"
done
