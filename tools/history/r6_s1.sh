#!/bin/bash
# round 6, GPU session 1: where the config-5 step spends its host time + PMC evidence for the volume kernels
REPO=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $REPO/gpurun_out/s1
cd $REPO
timeout 400 python tools/c5_profile.py > gpurun_out/s1/c5_host_profile.txt 2>&1; echo "host profile rc=$?"
head -60 gpurun_out/s1/c5_host_profile.txt | cut -c1-160
bash tools/r6_c5_pmc.sh 64,2048,2048 2>&1 | cut -c1-400
