"""host <-> device transfer rates through the library's own calls (pageable / page-locked), one thread"""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyimsegm_amd import _hip
from pyimsegm_amd.utilities.synthetic import voronoi_image
H = W = 2048
img = voronoi_image(H, W, seed=1)
pin = _hip.pinned_empty(img.shape, img.dtype); pin[...] = img
ctx = _hip.default_context()
sess = _hip.Image2D(H, W)
def t(fn, n=20):
    fn(); ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    ctx.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
print('upload pageable 12.6 MB: %.3f ms' % t(lambda: sess.upload(img)))
print('upload pinned   12.6 MB: %.3f ms' % t(lambda: sess.upload(pin)))
import ctypes as C
dev = C.c_void_p(); _hip._check(_hip.load_library().imsegm_device_alloc(0, 128 << 20, C.byref(dev)))
out_pin = _hip.pinned_empty((H, W), np.int32); out_pg = np.empty((H, W), np.int32)
print('D2H pinned   16.8 MB: %.3f ms' % t(lambda: ctx.copy(out_pin.ctypes.data, dev.value, out_pin.nbytes)))
print('D2H pageable 16.8 MB: %.3f ms' % t(lambda: ctx.copy(out_pg.ctypes.data, dev.value, out_pg.nbytes)))
soft = _hip.pinned_empty((H, W, 3), np.float64)
print('D2H pinned  100.7 MB: %.3f ms' % t(lambda: ctx.copy(soft.ctypes.data, dev.value, soft.nbytes), 5))
t0 = time.perf_counter(); a = _hip.pinned_empty((H, W, 5), np.float64); print('first pinned alloc 168 MB: %.3f ms' % ((time.perf_counter() - t0) * 1e3))
t0 = time.perf_counter(); b = np.array(img); print('numpy copy 12.6 MB: %.3f ms' % ((time.perf_counter() - t0) * 1e3))
t0 = time.perf_counter(); pin[...] = img; print('numpy copy into pinned 12.6 MB: %.3f ms' % ((time.perf_counter() - t0) * 1e3))
