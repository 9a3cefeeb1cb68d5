#!/bin/bash
# round 6, session 57: the raster-order update in workgroups of 512 and 1 024 lanes
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
IMSEGM_HIP_LIBRARY=$REPO/pyimsegm_amd/build/variants/upd512.so bash tools/c5_kstats.sh | grep "update_f32"
IMSEGM_HIP_LIBRARY=$REPO/pyimsegm_amd/build/variants/upd1024.so bash tools/c5_kstats.sh | grep "update_f32"
