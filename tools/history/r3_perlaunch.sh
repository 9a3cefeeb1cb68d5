#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
for g in ${GROUPS_LIST:-9 3 1}; do
  echo -n "sweeps per launch $g: "
  IMSEGM_SLIC_PERSISTENT=1 IMSEGM_SWEEPS_PER_LAUNCH=$g timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-other-configs --inflight 1 2>/dev/null | python tools/bl.py
  echo -n "sweeps per launch $g, 3 in flight: "
  IMSEGM_SLIC_PERSISTENT=1 IMSEGM_SWEEPS_PER_LAUNCH=$g timeout 200 python bench.py --steps 40 --warmup 3 --no-cpu-baseline --no-other-configs 2>/dev/null | python tools/bl.py
done
echo -n "per-sweep launches (round 2 path): "
timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-other-configs --inflight 1 2>/dev/null | python tools/bl.py
echo -n "per-sweep launches, 3 in flight: "
timeout 200 python bench.py --steps 40 --warmup 3 --no-cpu-baseline --no-other-configs 2>/dev/null | python tools/bl.py
