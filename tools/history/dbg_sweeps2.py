import os, sys, time, numpy as np
os.environ.setdefault('IMSEGM_SLIC_PERSISTENT', '1')
sys.path.insert(0, '/root/repo')
from pyimsegm_amd.superpixels import segment_slic_img2d
from pyimsegm_amd.utilities.synthetic import voronoi_image
from pyimsegm_amd import _hip
im = voronoi_image(2048, 2048, seed=1)
for i in range(3):
    t = time.perf_counter(); lab = segment_slic_img2d(im, 46, 0.2); print('ms', (time.perf_counter() - t) * 1e3, _hip.slic_sweep_runs(), flush=True)
