#!/bin/bash
# run selected GPU tests with a tight timeout: tools/t.sh <timeout_s> <pytest args...>
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
T=$1; shift
timeout $T python -m pytest -m gpu -x -q "$@" 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl" | tail -25
