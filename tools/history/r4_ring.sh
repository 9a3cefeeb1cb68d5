#!/bin/bash
# round 4: pageable input handed to the runtime vs the worker's own page-locked ring vs images that already live in page-locked memory
out=gpurun_out/$1; mkdir -p $out
for cfg in 4 2; do
  extra=""; [ $cfg = 2 ] && extra="--no-other-configs --steps 60"
  python bench.py --config $cfg --no-cpu-baseline $extra > $out/c${cfg}_pageable.json 2>/dev/null
  python bench.py --config $cfg --no-cpu-baseline $extra --input-ring 1 > $out/c${cfg}_ring.json 2>/dev/null
  python bench.py --config $cfg --no-cpu-baseline $extra --pinned-input 1 > $out/c${cfg}_pinned.json 2>/dev/null
done
python bench.py --gpus 2 --config 4 --steps 12 --no-cpu-baseline > $out/c4_gpus2.json 2> $out/c4_gpus2.err
python - <<PY
import json, glob
for f in sorted(glob.glob('$out/*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], d['value'], d['ms_per_step'], d['config']['input_memory'][:40], d.get('per_rank'), d.get('gather_backend'), d.get('gathered_maps_equal_reference_run'))
    except Exception as e:
        print(f, 'ERR', e)
PY
tail -c 400 $out/c4_gpus2.err
