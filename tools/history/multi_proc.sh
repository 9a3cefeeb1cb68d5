# throughput of P concurrent single-stream bench processes on one GPU (vs worker threads in one process)
cd ${GRAFT_REPO_ROOT:-/root/repo}
P=${1:-3}
for i in $(seq 1 $P); do
  python bench.py --steps 40 --warmup 3 --no-cpu-baseline --inflight 1 > /tmp/mp_$i.json 2>/dev/null &
done
wait
python - $P <<'PY'
import json, sys
tot = 0
for i in range(1, int(sys.argv[1]) + 1):
    d = json.loads(open('/tmp/mp_%d.json' % i).read().strip().splitlines()[-1])
    print('proc', i, 'ms/step', d['ms_per_step'], 'Mpx/s', d['value'])
    tot += d['value']
print('sum Mpx/s', round(tot, 1))
PY
