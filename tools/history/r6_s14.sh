#!/bin/bash
# sequential config-5 calls: current library against the volume kernels of commit 84e2f3e (same box, alternating)
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
for v in cur old cur old; do
  lib=$PWD/pyimsegm_amd/libimsegm_hip.so; [ $v = old ] && lib=$PWD/pyimsegm_amd/build/variants/oldvol.so
  IMSEGM_HIP_LIBRARY=$lib timeout 300 python tools/c5_steps.py 64,4096,4096 3 2>&1 | tail -4
done
