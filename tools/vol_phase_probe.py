"""where the 3-D float32 assignment kernel spends its time: shader-clock ticks per section, summed over the waves
(library variant built with -DIMSEGM_VOL_PHASE_PROF: tools/build_variant.sh volprof volume.hip -DIMSEGM_VOL_PHASE_PROF)

    IMSEGM_HIP_LIBRARY=$PWD/pyimsegm_amd/build/variants/volprof.so python tools/vol_phase_probe.py [D,H,W]
"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import bench
from pyimsegm_amd import _hip
from pyimsegm_amd import superpixels as S
from pyimsegm_amd.utilities.synthetic import config5_volume

shape = tuple(int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else '64,1024,1024').split(','))
vol = config5_volume(shape, seed=5)
p = bench.C5_PARAMS
lib = _hip.load_library()
read = lib.imsegm_debug_vol_phases
read.restype, read.argtypes = C.c_int, [C.POINTER(C.c_ulonglong), C.c_int]
sess = S._open_volume(vol)
S._run_slic3d(sess, p['sp_size'], p['sp_regul'], p['spacing'])          # warm-up
buf = (C.c_ulonglong * 16)()
read(buf, 1)
S._run_slic3d(sess, p['sp_size'], p['sp_regul'], p['spacing'])
read(buf, 1)
v = [int(x) for x in buf]
names = ['prologue + voxel loads', 'brick list scan + compaction', 'records into LDS', 'walk (bounds, selection, distances)',
         'end-of-batch barrier', 'label write', 'bounding boxes by runs']
waves = max(v[8], 1)
total = sum(v[:7])
print('volume %r, K = %d; %d waves over 10 sweeps' % (shape, sess.n_labels, waves))
for i, name in enumerate(names):
    print('%-40s %8.0f ticks per wave  %5.1f %%' % (name, v[i] / waves, 100.0 * v[i] / max(total, 1)))
print('%-40s %8.0f ticks per wave' % ('sum', total / waves))
print('brick list entries scanned per wave: %.1f;  staged candidates per batch-walk: %.1f;  candidates evaluated per wave: %.2f'
      % (v[9] / waves, v[10] / waves, v[11] / waves))
sess.close()
