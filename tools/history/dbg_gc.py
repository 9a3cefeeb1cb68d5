import os, sys, numpy as np
sys.path.insert(0, '/root/repo')
import bench
from pyimsegm_amd import pipelines as pipe
from pyimsegm_amd.descriptors import FEATURES_SET_COLOR
from pyimsegm_amd.utilities.synthetic import voronoi_image
ref = bench.load_golden('reference_c4.npz'); model = bench.model_from_arrays(ref)
for seed in (100, 101, 102):
    im = voronoi_image(647, 1024, seed=seed)
    pipe._segment_color2d_one_call(im, model, FEATURES_SET_COLOR, 35, 0.2, 2.0, 'model', want_soft=False, reuse=True)
ref2 = bench.load_golden('reference_2048.npz'); model2 = bench.model_from_arrays(ref2)
im = voronoi_image(2048, 2048, seed=1)
for _ in range(3):
    pipe._segment_color2d_one_call(im, model2, FEATURES_SET_COLOR, 46, 0.2, 2.0, 'model', want_soft=False, reuse=True)
