"""D2H / H2D rates with several threads (one context = one stream each) copying at the same time: is the 2048^2 line bound by PCIe?"""
import sys, os, time, threading
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyimsegm_amd import _hip
import ctypes as C
H = W = 2048
def worker(n_iter, direction, results, i, barrier):
    ctx = _hip.default_context()
    dev = C.c_void_p(); _hip._check(_hip.load_library().imsegm_device_alloc(0, 32 << 20, C.byref(dev)))
    out_pin = _hip.pinned_empty((H, W), np.int32)
    in_pg = np.zeros((H, W, 3), np.uint8)
    def d2h(): ctx.copy(out_pin.ctypes.data, dev.value, out_pin.nbytes)
    def h2d(): ctx.copy(dev.value, in_pg.ctypes.data, in_pg.nbytes)
    fns = {'d2h': [d2h], 'h2d': [h2d], 'both': [h2d, d2h]}[direction]
    for f in fns: f()
    ctx.synchronize()
    barrier.wait()
    t0 = time.perf_counter()
    for _ in range(n_iter):
        for f in fns: f()
        ctx.synchronize()
    results[i] = (time.perf_counter() - t0) / n_iter
for direction in ('d2h', 'h2d', 'both'):
    for nthreads in (1, 2, 4, 6):
        res = [0] * nthreads
        bar = threading.Barrier(nthreads)
        th = [threading.Thread(target=worker, args=(30, direction, res, i, bar)) for i in range(nthreads)]
        [t.start() for t in th]; [t.join() for t in th]
        per = max(res)
        mb = {'d2h': 16.8, 'h2d': 12.6, 'both': 29.4}[direction]
        print('%-4s %d threads: %.3f ms per round per thread, aggregate %.1f GB/s, %.0f images/s' % (direction, nthreads, per * 1e3, nthreads * mb / per / 1e3, nthreads / per))
