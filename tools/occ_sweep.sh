cd ${GRAFT_REPO_ROOT:-/root/repo}
for lds in 0 38000 50000 78000 150000; do
  IMSEGM_ASSIGN_LDS=$lds python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('lds=$lds', 'assign_us', d['roofline']['avg_kernel_us'], 'ms/step', d['ms_per_step'])"
done
