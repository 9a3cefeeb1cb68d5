"""The host's mixture fit of config 5 on ITS OWN feature table (one pipeline call at BASELINE size captures it), alone, against the
budget of BLAS callers the side-by-side restarts may use (graph_cuts._SIDE_BY_SIDE.callers: 28 = two fits at once stay below 64).

    python tools/fit_team_probe.py
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from pyimsegm_amd import graph_cuts as G  # noqa: E402
from pyimsegm_amd import pipelines as P  # noqa: E402
from pyimsegm_amd.utilities.synthetic import config5_volume  # noqa: E402

shape = tuple(int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else '64,4096,4096').split(','))
vol = config5_volume(shape, seed=5)
seen = {}
real = P.estim_class_model


def recording(features, nb_classes, *a, **kw):
    seen['features'], seen['nb'] = np.array(features), nb_classes
    return real(features, nb_classes, *a, **kw)


P.estim_class_model = recording
p = bench.C5_PARAMS
np.random.seed(0)
P.pipe_gray3d_slic_features_model_graphcut(vol, bench.NB_CLASSES, {'color': ('mean', 'std', 'energy')}, spacing=p['spacing'], sp_size=p['sp_size'],
                                           sp_regul=p['sp_regul'], gc_regul=p['gc_regul'])
feats, nb = seen['features'], seen['nb']
print('feature table', feats.shape, 'classes', nb)
for callers in (28, 56, 28, 56, 19):
    G._SIDE_BY_SIDE.callers = callers
    times = []
    for rep in range(4):
        np.random.seed(rep)
        t0 = time.perf_counter()
        G.estim_class_model(feats, nb)
        times.append(time.perf_counter() - t0)
    print('callers %2d: %s' % (callers, ' '.join('%.3f' % t for t in times)))
