"""cProfile of the host side of the benchmark step (single thread, GPU box)."""
import cProfile
import os
import pstats
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyimsegm_amd import graph_cuts as G  # noqa: E402
from pyimsegm_amd import pipelines as pipe  # noqa: E402
from pyimsegm_amd.descriptors import FEATURES_SET_COLOR  # noqa: E402
from pyimsegm_amd.superpixels import _open_session  # noqa: E402
from pyimsegm_amd.utilities.synthetic import voronoi_image  # noqa: E402

img = voronoi_image(2048, 2048)
session = _open_session(img)
res0 = pipe._ResidentImage(img, FEATURES_SET_COLOR, 46, 0.2, session=session)
np.random.seed(0)
model = G.estim_class_model(res0.features, 3, 'GMM', None, True)


def step():
    res = pipe._ResidentImage(img, FEATURES_SET_COLOR, 46, 0.2, session=session)
    proba = model.predict_proba(res.features)
    return res.segment(proba, 2.0, 'model', to_host=False)


for _ in range(5):
    step()
pr = cProfile.Profile()
pr.enable()
for _ in range(100):
    step()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats('tottime').print_stats(28)
