#!/bin/bash
# round 6, session 54: waves per SIMD of the tiled assignment: 6 against 5
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
IMSEGM_HIP_LIBRARY=$REPO/pyimsegm_amd/build/variants/waves6.so bash tools/c5_kstats.sh | grep "k_vol_assign"
IMSEGM_HIP_LIBRARY=$REPO/pyimsegm_amd/build/variants/waves5.so bash tools/c5_kstats.sh | grep "k_vol_assign"
