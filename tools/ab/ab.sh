#!/bin/bash
# same box, alternating: old build / new build of the library on the config-2 line (and the config-4 line one image at a time)
out=gpurun_out/$1; mkdir -p $out
for i in 1 2; do
  IMSEGM_HIP_LIBRARY=$PWD/tools/ab/libimsegm_hip_old.so python tools/ab/run.py --no-other-configs --no-cpu-baseline --steps 60 > $out/old_c2_$i.json 2>/dev/null
  python tools/ab/run.py --no-other-configs --no-cpu-baseline --steps 60 > $out/new_c2_$i.json 2>/dev/null
done
IMSEGM_HIP_LIBRARY=$PWD/tools/ab/libimsegm_hip_old.so python tools/ab/run.py --config 4 --no-cpu-baseline --batch-images 0 > $out/old_c4.json 2>/dev/null
python tools/ab/run.py --config 4 --no-cpu-baseline --batch-images 0 > $out/new_c4.json 2>/dev/null
python tools/ab/run.py --config 4 --no-cpu-baseline > $out/new_c4_batch.json 2>/dev/null
python - <<PY
import json, glob
for f in sorted(glob.glob('$out/*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], d['value'], d['roofline'].get('avg_kernel_us'), d.get('device_resident', {}).get('ms_per_step'), d['stage_ms_per_step'])
    except Exception as e:
        print(f, 'ERR', e)
PY
