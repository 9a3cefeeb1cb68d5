"""wall-clock timeline of one config-5 call: every stage of pipelines._gray3d_on_session with its start and duration (the bench's
thread limit for the fit applied), three calls

    python tools/c5_timeline.py [D,H,W]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from threadpoolctl import threadpool_limits

import bench
from pyimsegm_amd import _hip
from pyimsegm_amd import pipelines as pipe
from pyimsegm_amd import superpixels as S
from pyimsegm_amd.utilities.synthetic import config5_volume

shape = tuple(int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else '64,4096,4096').split(','))
vol = config5_volume(shape, seed=5)
p = bench.C5_PARAMS
limiter = threadpool_limits(limits=min(32, os.cpu_count() or 1))
T0 = [0.0]
LOG = []


def timed(name, fn):
    def wrapper(*a, **kw):
        t = time.perf_counter()
        try:
            return fn(*a, **kw)
        finally:
            LOG.append((name, t - T0[0], time.perf_counter() - t))
    return wrapper


pipe._open_volume = timed('open + upload', pipe._open_volume)
pipe._run_slic3d = timed('slic + label_cc', pipe._run_slic3d)
pipe.compute_selected_features_gray3d = timed('features', pipe.compute_selected_features_gray3d)
pipe.norm_features = timed('norm_features', pipe.norm_features)
pipe.estim_class_model = timed('fit', pipe.estim_class_model)
pipe.predict_proba = timed('predict_proba', pipe.predict_proba)
_hip.Volume3D.graph_prepare = timed('graph_prepare (enqueue)', _hip.Volume3D.graph_prepare)
_hip.Volume3D.segment = timed('segment (terms, cut, gather, download)', _hip.Volume3D.segment)
touched = pipe._touched_result


def touched_timed(*a, **kw):
    out, join = touched(*a, **kw)
    return out, timed('join page-touching threads', join)


pipe._touched_result = timed('result array + touch threads start', touched_timed)
for i in range(3):
    np.random.seed(0)
    del LOG[:]
    T0[0] = time.perf_counter()
    segm = pipe.pipe_gray3d_slic_features_model_graphcut(vol, bench.NB_CLASSES, {'color': ('mean', 'std', 'energy')}, spacing=p['spacing'],
                                                         sp_size=p['sp_size'], sp_regul=p['sp_regul'], gc_regul=p['gc_regul'])
    total = time.perf_counter() - T0[0]
    print('call %d: %.3f s' % (i, total))
    for name, start, dur in LOG:
        print('    %-42s at %7.1f ms  %7.1f ms' % (name, start * 1e3, dur * 1e3))
    t = time.perf_counter()
    del segm
    print('    (freeing the result: %.1f ms)' % ((time.perf_counter() - t) * 1e3))
