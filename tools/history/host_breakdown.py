#!/usr/bin/env python
"""Wall-clock breakdown of one bench step on the host side (each stage followed by a device sync)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyimsegm_amd import _hip, graph_cuts as G  # noqa: E402
from pyimsegm_amd import pipelines as pipe  # noqa: E402
from pyimsegm_amd.descriptors import FEATURES_SET_COLOR, _selected_features_color2d  # noqa: E402
from pyimsegm_amd.superpixels import _open_session, _run_slic  # noqa: E402
from pyimsegm_amd.utilities.synthetic import voronoi_image  # noqa: E402

img = voronoi_image(2048, 2048)
ctx = _hip.default_context()
sess, mode = _open_session(img)
res0 = pipe._ResidentImage(img, FEATURES_SET_COLOR, 46, 0.2, session=(sess, mode))
np.random.seed(0)
model = G.estim_class_model(res0.features, 3, 'GMM', None, True)
acc = {}


def tic(name, t0):
    ctx.synchronize()
    t1 = time.perf_counter()
    acc[name] = acc.get(name, 0.0) + (t1 - t0)
    return t1


N = 30
for it in range(N + 3):
    if it == 3:
        acc.clear()
    t = time.perf_counter()
    _run_slic(sess, mode, 46, 0.2); t = tic('slic', t)
    fts, _ = _selected_features_color2d(img, None, FEATURES_SET_COLOR, sess=sess); t = tic('stats', t)
    proba = model.predict_proba(fts); t = tic('predict_proba', t)
    edges, centres, present = sess.graph(); t = tic('graph', t)
    edges = np.array(edges, dtype=np.int32).reshape(-1, 2)
    w = G.compute_edge_model(edges, proba, 'lT')
    w = w / G.compute_spatial_dist(centres, edges, relative=True)
    w = np.clip(w, 1e-3, 1e3)
    unary = G.compute_unary_cost(proba)
    pw = G.compute_pairwise_cost(2.0, proba.shape); t = tic('edge_weights+unary', t)
    labels = _hip.cut_general_graph(edges, w, unary, pw); t = tic('graphcut', t)
    sess.gather(labels, proba, to_host=False); t = tic('gather', t)
tot = sum(acc.values())
for k, v in acc.items():
    print('%-20s %7.3f ms' % (k, v / N * 1e3))
print('%-20s %7.3f ms' % ('total', tot / N * 1e3))
