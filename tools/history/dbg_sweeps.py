import os, sys, numpy as np
os.environ.setdefault('IMSEGM_SLIC_PERSISTENT', '1')
sys.path.insert(0, '/root/repo')
from pyimsegm_amd.superpixels import segment_slic_img2d
from pyimsegm_amd.utilities.synthetic import voronoi_image
from pyimsegm_amd import _hip
for shape, sp, rc, seed in [((512, 640), 30, 0.2, 3), ((2048, 2048), 46, 0.2, 1), ((647, 1024), 35, 0.2, 100)]:
    im = voronoi_image(shape[0], shape[1], seed=seed)
    lab = segment_slic_img2d(im, sp, rc)
    print(shape, _hip.slic_sweep_runs(), int(lab.max()) + 1, flush=True)
rng = np.random.default_rng(3)
im = rng.random((97, 131, 3))
lab = segment_slic_img2d(im, 11, 0.1)
print('float', _hip.slic_sweep_runs(), flush=True)
