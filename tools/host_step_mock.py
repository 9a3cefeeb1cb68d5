"""Host share of a pipeline step without a GPU: the session's C calls are replaced by canned results of the benchmark's
shape (K = 2025 superpixels, E = 5900 edges), so what remains is the Python / numpy work of the calling thread plus the
round trip to the helper processes (pyimsegm_amd/hostpool.py).  Prints the step rate for 1 / 3 / 6 worker threads and a
profile of the calling thread.

    python tools/host_step_mock.py [ms of simulated kernel time per step]
"""
import sys, time, cProfile, pstats, threading
import numpy as np
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from threadpoolctl import threadpool_limits
threadpool_limits(1)
from pyimsegm_amd import _hip, pipelines as pipe, graph_cuts as G
from pyimsegm_amd.descriptors import FEATURES_SET_COLOR
from pyimsegm_amd.hostpool import HostMathPool
K, E = 2025, 5900
rng = np.random.default_rng(0)
fts3 = [rng.random((K, 3)) * s for s in (200., 30., 40000.)]
edges = np.stack([rng.integers(0, K - 3, E), np.zeros(E, int)], 1)
edges[:, 1] = edges[:, 0] + 1 + rng.integers(0, 2, E)
edges = edges.astype(np.int32)
centres = rng.random((K, 2)) * 2048
GPU_MS = float(sys.argv[1]) if len(sys.argv) > 1 else 0.0
class Fake(object):
    shape = (2048, 2048); n_labels = K; ctx = None
    def slic(self, *a, **k):
        if GPU_MS: time.sleep(GPU_MS * 1e-3)      # GIL released, like the C call
        return K
    def color_stats(self, mean=True, energy=True, var=True): return fts3[0].copy(), fts3[2].copy(), fts3[1].copy() ** 2
    def graph(self, *a, **k): return edges.copy(), centres.copy(), np.ones(K, np.uint8)
    def gather(self, gl, pr, to_host=True): return None, None
    def close(self): pass
image = np.zeros((2048, 2048, 3), np.uint8)
res0 = pipe._ResidentImage(image, FEATURES_SET_COLOR, 46, 0.2, session=(Fake(), 2))
np.random.seed(0)
model = G.estim_class_model(res0.features, 3, 'GMM', None, True)
_hip.cut_general_graph = lambda e, w, u, p, **k: np.zeros(K, np.int32)
pool = HostMathPool(6); pool.set_model(model)
def step(sess):
    res = pipe._ResidentImage(image, FEATURES_SET_COLOR, 46, 0.2, session=(sess, 2))
    return res.segment_with_model(model, 2.0, 'model', pool, to_host=False)
s0 = Fake(); step(s0)
for nt in (1, 3, 6):
    def work(n):
        s = Fake()
        for _ in range(n): step(s)
    ths = [threading.Thread(target=work, args=(60,)) for _ in range(nt)]
    t0 = time.perf_counter(); [t.start() for t in ths]; [t.join() for t in ths]
    print('%d threads: %.0f us per step' % (nt, (time.perf_counter() - t0) / (60 * nt) * 1e6))
pr = cProfile.Profile(); pr.enable()
for _ in range(100): step(s0)
pr.disable()
pstats.Stats(pr).sort_stats('tottime').print_stats(14)
pool.close()
