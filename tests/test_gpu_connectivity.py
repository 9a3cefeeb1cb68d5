"""``_enforce_label_connectivity_cython`` (scikit-image 0.18, the second native call inside ``skimage.segmentation.slic``;
/root/reference/imsegm/superpixels.py:61-63) on crafted label maps, HIP against the oracle, bit for bit.

The 2-D tile path of ``csrc/connectivity.hip`` has hand-over points (more than 64 local components in a tile, a BFS
frontier of more than 64 cells, bounding boxes beyond the LDS tile, oversize components); every case below is built to
cross one of them.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def hip():
    from pyimsegm_amd import _hip
    _hip.default_context()
    return _hip


def _blocks(h, w, bh, bw):
    yy, xx = np.mgrid[0:h, 0:w]
    return ((yy // bh) * ((w + bw - 1) // bw) + xx // bw).astype(np.int32)


def _salted(h, w, bh, bw, frac, seed):
    rng = np.random.RandomState(seed)
    lab = _blocks(h, w, bh, bw)
    m = rng.rand(h, w) < frac
    lab[m] = rng.randint(0, lab.max() + 1, m.sum())
    return lab


def _comb(h, w):
    """one block label with a comb of another label inside it: small component with a wide BFS frontier"""
    lab = np.zeros((h, w), np.int32)
    lab[:, w // 2:] = 1
    lab[4, 4:4 + 150] = 2                     # spine ...
    for x in range(4, 4 + 150, 2):
        lab[5:12, x] = 2                      # ... and 75 teeth: the frontier grows to 75 cells
    return lab


def _diagonal(h, w, n):
    lab = _blocks(h, w, 64, 64)
    for i in range(n):                        # one-pixel staircase of a foreign label: thin, bounding box n x n
        lab[10 + i, 10 + i] = 999
        lab[10 + i, 11 + i] = 999
    return lab


CASES = [
    ('blocks_ragged', lambda: _blocks(203, 317, 23, 31), 100, 2000),
    ('salt_1pct', lambda: _salted(256, 320, 32, 40, 0.01, 0), 300, 5000),
    ('salt_10pct', lambda: _salted(200, 200, 25, 25, 0.10, 1), 200, 3000),
    ('noise', lambda: np.random.RandomState(2).randint(0, 6, (150, 170)).astype(np.int32), 20, 400),
    ('comb_frontier_75', lambda: _comb(64, 400), 1000, 100000),
    ('diagonal_100', lambda: _diagonal(256, 256, 100), 500, 100000),
    ('diagonal_200', lambda: _diagonal(320, 320, 200), 500, 100000),
    ('oversize', lambda: _blocks(128, 128, 64, 64), 10, 1000),
    ('one_row', lambda: _blocks(1, 500, 1, 37), 20, 100),
    ('one_column', lambda: _blocks(500, 1, 41, 1), 20, 100),
    ('tiny', lambda: _blocks(4, 4, 2, 2), 2, 100),
    ('everything_small', lambda: _salted(96, 96, 8, 8, 0.2, 3), 100000, 1000000),
]


#: which path has to finish the case: the tile path (incl. its own hand-overs to the 128 K tile and the LDS-frontier kernel), or
#: the general one (None: either)
PATH = {'blocks_ragged': 'tile', 'salt_1pct': 'tile', 'comb_frontier_75': 'tile', 'diagonal_100': 'tile', 'diagonal_200': 'tile',
        'one_row': 'tile', 'one_column': 'tile', 'tiny': 'tile', 'noise': 'general', 'oversize': 'general'}


@pytest.mark.parametrize('name,make,min_size,max_size', CASES, ids=[c[0] for c in CASES])
@pytest.mark.parametrize('start_label', [0, 1])
def test_enforce_connectivity_bit_exact(hip, oracle, name, make, min_size, max_size, start_label):
    lab = make() + start_label          # (as slic hands it over: no pixel carries the mask label start_label - 1)
    want = oracle.enforce_connectivity(lab, min_size, max_size, start_label)
    im = hip.Image2D(lab.shape[0], lab.shape[1])
    general_runs = hip.load_library().imsegm_debug_conn_general_runs
    try:
        before = general_runs()
        got = im.enforce_connectivity(lab, min_size, max_size, start_label)
        took_general = general_runs() - before
        assert got.shape == lab.shape and np.array_equal(got, want), '%d pixels differ' % int((got != want).sum())
        if PATH.get(name) == 'tile':
            assert took_general == 0, 'left the tile path'
        elif PATH.get(name) == 'general':
            assert took_general == 1
        # session reuse: a second, different map on the same buffers
        lab2 = np.ascontiguousarray(lab[::-1])
        assert np.array_equal(im.enforce_connectivity(lab2, min_size, max_size, start_label),
                              oracle.enforce_connectivity(lab2, min_size, max_size, start_label))
    finally:
        im.close()


def test_enforce_connectivity_randomised(hip, oracle):
    rng = np.random.RandomState(7)
    for _ in range(12):
        h, w = int(rng.randint(5, 300)), int(rng.randint(5, 300))
        bh, bw = int(rng.randint(3, 40)), int(rng.randint(3, 40))
        lab = _salted(h, w, bh, bw, float(rng.choice([0.0, 0.002, 0.02, 0.1])), int(rng.randint(1 << 30)))
        seg = h * w / float(lab.max() + 1)
        min_size, max_size = int(0.5 * seg), int(3 * seg)
        im = hip.Image2D(h, w)
        try:
            got = im.enforce_connectivity(lab, min_size, max_size, 0)
        finally:
            im.close()
        assert np.array_equal(got, oracle.enforce_connectivity(lab, min_size, max_size, 0)), (h, w, bh, bw)


def test_enforce_connectivity_volume(hip, oracle):
    """3-D label maps (6 neighbours, z first) take the general path through the same entry point"""
    rng = np.random.RandomState(11)
    for shape, b in (((6, 40, 50), 10), ((12, 33, 21), 7), ((1, 64, 64), 16)):
        zz, yy, xx = np.mgrid[0:shape[0], 0:shape[1], 0:shape[2]]
        lab = ((zz // max(1, shape[0] // 2)) * 100 + (yy // b) * 10 + xx // b).astype(np.int32)
        m = rng.rand(*shape) < 0.03
        lab[m] = rng.randint(0, 40, m.sum())
        seg = lab.size / float(len(np.unique(lab)))
        want = oracle.enforce_connectivity(lab, int(0.5 * seg), int(3 * seg), 0)
        vol = hip.Volume3D(*shape)
        try:
            got = vol.enforce_connectivity(lab, int(0.5 * seg), int(3 * seg), 0)
        finally:
            vol.close()
        assert np.array_equal(got, want), shape


def test_slic_of_benchmark_like_images_stays_on_the_tile_path(hip):
    """SLIC label maps of the synthetic benchmark images (configs 2 and 4) must not need the general connectivity path"""
    from pyimsegm_amd import superpixels as S
    from pyimsegm_amd.utilities.synthetic import voronoi_image
    general_runs = hip.load_library().imsegm_debug_conn_general_runs
    before = general_runs()
    for shape, sp, seed in (((647, 1024), 35, 100), ((647, 1024), 35, 101), ((1024, 1024), 46, 1)):
        S.segment_slic_img2d(voronoi_image(shape[0], shape[1], seed=seed), sp, 0.2)
    assert general_runs() == before
