#!/usr/bin/env python
"""Time BASELINE config 3 (2048 x 2048, SLIC + full Leung-Malik bank statistics) on the GPU and one
battery of the reference's scipy formulation on a crop for scale.  python tools/bench_texture.py [size]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyimsegm_amd import _hip  # noqa: E402
from pyimsegm_amd import descriptors as D  # noqa: E402
from pyimsegm_amd.superpixels import _open_session, _run_slic  # noqa: E402
from pyimsegm_amd.utilities.synthetic import voronoi_image  # noqa: E402

size = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
img = voronoi_image(size, size)
sess, mode = _open_session(img)
_run_slic(sess, mode, 46, 0.2)
filters, names = D._select_bank('normal')
ctx = _hip.default_context()
for rep in range(2):
    ctx.synchronize()
    t0 = time.perf_counter()
    fts, ns = D._texture_desc_lm_device(img, None, ('mean', 'std', 'energy'), filters, names, sess=sess)
    ctx.synchronize()
    dt = time.perf_counter() - t0
print('GPU: tLM mean/std/energy on %dx%d, K=%d -> features %r in %.1f ms (%.1f Mpx/s)' %
      (size, size, sess.n_labels, fts.shape, dt * 1e3, size * size / dt / 1e6))
flops = 2.0 * 33 * 33 * 76 * 3 * size * size
print('     filter bank: %.2f TFLOP fp64 -> %.1f TFLOP/s incl. high-pass, norms, statistics' % (flops / 1e12, flops / dt / 1e12))
# CPU scale: one 8-kernel battery of the reference formulation on a 256 x 256 crop, one channel
from scipy import ndimage  # noqa: E402
crop = img[:256, :256, 0].astype(float)
t0 = time.perf_counter()
_ = [ndimage.convolve(crop, fl) for fl in filters[0]]
dt_cpu = time.perf_counter() - t0
full = dt_cpu / 8 * 76 * 3 * (size * size) / (256 * 256)
print('CPU: scipy.ndimage.convolve 8 kernels on 256x256x1 in %.2f s -> extrapolated full bank at %dx%dx3: %.0f s' %
      (dt_cpu, size, size, full))
