#!/bin/bash
# A/B of library variants x workgroups per CU of the persistent sweep kernel: HIP-event average of the kernel (bench.py, one image in flight)
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
export IMSEGM_SLIC_PERSISTENT=1
cp pyimsegm_amd/libimsegm_hip.so /tmp/lib_orig.so
for v in "$@"; do
  cp pyimsegm_amd/build/variants/$v.so pyimsegm_amd/libimsegm_hip.so
  for b in ${BLOCKS:-5}; do
    echo -n "$v blocks/CU=$b: "
    IMSEGM_SWEEPS_BLOCKS_PER_CU=$b timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-other-configs --inflight 1 2>/dev/null | python tools/bl.py
  done
done
cp /tmp/lib_orig.so pyimsegm_amd/libimsegm_hip.so
