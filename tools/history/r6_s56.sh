#!/bin/bash
# round 6, session 56: the raster-order update in workgroups of one wave (variant upd64) against four
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
bash tools/c5_kstats.sh | grep "update_f32"
IMSEGM_HIP_LIBRARY=$REPO/pyimsegm_amd/build/variants/upd64.so bash tools/c5_kstats.sh | grep "update_f32"
