#!/usr/bin/env python
"""Golden vectors of ``skimage.segmentation._slic._enforce_label_connectivity_cython`` -- the REAL scikit-image 0.18.3 (the
connectivity pass inside ``skimage.segmentation.slic``, reached from imsegm/superpixels.py:61-63 and :104-106) -- on crafted
label maps: the shapes ``tests/test_gpu_connectivity.py`` drives through the hand-over points of the HIP tile path (combs with
wide BFS frontiers, thin diagonals, salt noise, oversize components, one row / one column, a volume).  SLIC label maps are
covered by ``make_golden_skimage.py``; these are not SLIC outputs, so they pin the restatement where SLIC never goes.

    /opt/conda/bin/python3.9 tests/golden/make_golden_connectivity.py        ->  tests/golden/connectivity.npz

The generators are plain numpy with fixed seeds and are re-run by the tests (the inputs are stored as CRCs only).
"""
import os
import sys
import zlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def blocks(h, w, bh, bw):
    yy, xx = np.mgrid[0:h, 0:w]
    return ((yy // bh) * ((w + bw - 1) // bw) + xx // bw).astype(np.int32)


def salted(h, w, bh, bw, frac, seed):
    rng = np.random.RandomState(seed)
    lab = blocks(h, w, bh, bw)
    m = rng.rand(h, w) < frac
    lab[m] = rng.randint(0, lab.max() + 1, m.sum())
    return lab


def comb(h, w, x0, teeth, depth):
    lab = np.zeros((h, w), np.int32)
    lab[:, w // 2:] = 1
    lab[4, x0:x0 + 2 * teeth] = 2
    for x in range(x0, x0 + 2 * teeth, 2):
        lab[5:5 + depth, x] = 2
    return lab


def diagonal(h, w, n, b=64):
    lab = blocks(h, w, b, b)
    for i in range(n):
        lab[10 + i, 10 + i] = 999
        lab[10 + i, 11 + i] = 999
    return lab


def volume(d, h, w, b, frac, seed):
    rng = np.random.RandomState(seed)
    zz, yy, xx = np.mgrid[0:d, 0:h, 0:w]
    lab = ((zz // max(1, d // 2)) * 100 + (yy // b) * 10 + xx // b).astype(np.int32)
    m = rng.rand(d, h, w) < frac
    lab[m] = rng.randint(0, 40, m.sum())
    return lab


#: name -> (generator expression, min_size, max_size)
CASES = {
    'blocks_ragged': ('blocks(203, 317, 23, 31)', 100, 2000),
    'salt_1pct': ('salted(256, 320, 32, 40, 0.01, 0)', 300, 5000),
    'salt_10pct': ('salted(200, 200, 25, 25, 0.10, 1)', 200, 3000),
    'noise': ('np.random.RandomState(2).randint(0, 6, (150, 170)).astype(np.int32)', 20, 400),
    'comb_frontier_75': ('comb(64, 400, 4, 75, 7)', 1000, 100000),
    'diagonal_100': ('diagonal(256, 256, 100)', 500, 100000),
    'diagonal_200': ('diagonal(320, 320, 200)', 500, 100000),
    'oversize': ('blocks(128, 128, 64, 64)', 10, 1000),
    'one_row': ('blocks(1, 500, 1, 37)', 20, 100),
    'one_column': ('blocks(500, 1, 41, 1)', 20, 100),
    'everything_small': ('salted(96, 96, 8, 8, 0.2, 3)', 100000, 1000000),
    'volume': ('volume(6, 40, 50, 10, 0.03, 11)', 60, 400),
}


def make(name):
    return eval(CASES[name][0])     # noqa: S307 -- fixed expressions above


def main():
    import skimage
    from skimage.segmentation._slic import _enforce_label_connectivity_cython
    out = {'skimage_version': np.array(skimage.__version__)}
    for name, (_, min_size, max_size) in CASES.items():
        for start_label in (0, 1):
            lab = make(name) + start_label
            seg = (lab[np.newaxis] if lab.ndim == 2 else lab).astype(np.intp)
            res = np.asarray(_enforce_label_connectivity_cython(np.ascontiguousarray(seg), min_size, max_size, start_label=start_label))
            res = res[0] if lab.ndim == 2 else res
            key = '%s_start%d' % (name, start_label)
            out[key] = res.astype(np.int32)
            out[key + '_input_crc'] = np.array(zlib.crc32(np.ascontiguousarray(lab, dtype=np.int32).tobytes()))
            print(key, lab.shape, 'labels', int(res.max()) + 1)
    np.savez_compressed(os.path.join(HERE, 'connectivity.npz'), **out)


if __name__ == '__main__':
    sys.exit(main())
