#!/bin/bash
# Round-5 profile set: tools/profile_round4.sh with the tag r05 (the driver's command, kernel statistics of configs 2 / 3 / 4, PMC
# traffic + SQ counters of the assignment kernel, two ranks on one device) + kernel statistics of config 5 at BASELINE size.
# Summaries land in gpurun_out/prof_r05/ (+ gpurun_out/c5ks/) and are copied into profiles/ by hand.
REPO=${GRAFT_REPO_ROOT:-/root/repo}
bash $REPO/tools/profile_round4.sh r05
bash $REPO/tools/r4_c5_kstats.sh
