"""Randomised parity sweep of the 2-D SLIC path against the CPU oracle (GPU box; test infrastructure)."""
import sys
import time

import numpy as np

sys.path.insert(0, __file__.rsplit('/', 2)[0])
from oracle import oracle as O  # noqa: E402
from pyimsegm_amd import _hip  # noqa: E402
from pyimsegm_amd.utilities.synthetic import voronoi_image  # noqa: E402

rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
n_cases = int(sys.argv[2]) if len(sys.argv) > 2 else 24
bad = 0
t0 = time.time()
for case in range(n_cases):
    H, W = int(rng.integers(40, 420)), int(rng.integers(40, 520))
    kind = case % 4
    if kind == 0:
        img = voronoi_image(H, W, seed=int(rng.integers(1 << 30)))
    elif kind == 1:
        img = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)                      # pure noise
    elif kind == 2:
        img = rng.random((H, W, 3))                                               # float noise
    else:
        yy, xx = np.mgrid[:H, :W]
        img = np.stack([np.sin(yy / 7.0) * 0.5 + 0.5, np.cos(xx / 11.0) * 0.5 + 0.5, (yy + xx) / (H + W)], axis=-1)  # smooth
        img = (img * 255).astype(np.uint8) if case % 8 == 3 else img
    sp = int(rng.integers(5, 40))
    regul = float(rng.choice([0.02, 0.1, 0.2, 0.5, 1.0, 3.0]))
    n_seg = int(H * W / sp**2)
    compact = (sp * regul)**1.5
    if n_seg < 1:
        continue
    ref = O.segment_slic_img2d(img, sp, regul)
    sess = _hip.Image2D(H, W).upload(img)
    sess.slic(n_seg, compact, sigma=1., normalize=2)
    got = sess.get_labels()
    sess.close()
    ok = np.array_equal(got, ref)
    bad += not ok
    print('case %2d kind %d %3dx%3d sp %2d regul %.2f (compactness %.3f): %s' %
          (case, kind, H, W, sp, regul, compact, 'ok' if ok else 'MISMATCH in %d px' % np.count_nonzero(got != ref)), flush=True)
print('%d cases, %d mismatches, %.1f s' % (n_cases, bad, time.time() - t0))
sys.exit(1 if bad else 0)
