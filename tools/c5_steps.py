"""sequential config-5 volumes on the calling thread: seconds per call and the share of the host-side mixture fit (bench.py's thread
limit for the fit applied)

    python tools/c5_steps.py [D,H,W] [calls]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from threadpoolctl import threadpool_limits

import bench
from pyimsegm_amd import _hip
from pyimsegm_amd import pipelines as pipe
from pyimsegm_amd.utilities.synthetic import config5_volume

shape = tuple(int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else '64,4096,4096').split(','))
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 4
vol = config5_volume(shape, seed=5)
p = bench.C5_PARAMS
limiter = threadpool_limits(limits=min(32, os.cpu_count() or 1))
fit_s = [0.0]
fit = pipe.estim_class_model


def timed_fit(*a, **kw):
    t = time.perf_counter()
    try:
        return fit(*a, **kw)
    finally:
        fit_s[0] += time.perf_counter() - t


pipe.estim_class_model = timed_fit
for i in range(calls + 1):
    np.random.seed(0)
    fit_s[0] = 0.0
    t = time.perf_counter()
    segm = pipe.pipe_gray3d_slic_features_model_graphcut(vol, bench.NB_CLASSES, {'color': ('mean', 'std', 'energy')}, spacing=p['spacing'],
                                                         sp_size=p['sp_size'], sp_regul=p['sp_regul'], gc_regul=p['gc_regul'])
    dt = time.perf_counter() - t
    print('%s call %d: %.3f s, fit %.3f s, rest %.3f s' % (os.path.basename(_hip.LIB_PATH), i, dt, fit_s[0], dt - fit_s[0]), flush=True)
