#!/bin/bash
# round 6, session 53: waves per SIMD the register allocation of the tiled assignment aims at: 7 (library) against 6 and 8
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
bash tools/c5_kstats.sh | grep "k_vol_assign"
IMSEGM_HIP_LIBRARY=$REPO/pyimsegm_amd/build/variants/waves6.so bash tools/c5_kstats.sh | grep "k_vol_assign"
IMSEGM_HIP_LIBRARY=$REPO/pyimsegm_amd/build/variants/waves8.so bash tools/c5_kstats.sh | grep "k_vol_assign"
