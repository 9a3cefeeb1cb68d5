cd /root/repo
for dbg in 0 1 2 4 8 3 7; do
  IMSEGM_DEBUG_ASSIGN=$dbg python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('debug=$dbg', 'assign_us', d['roofline']['avg_kernel_us'], 'ms/step', d['ms_per_step'])"
done
