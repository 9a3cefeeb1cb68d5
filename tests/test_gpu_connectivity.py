"""``_enforce_label_connectivity_cython`` (scikit-image 0.18, the second native call inside ``skimage.segmentation.slic``;
/root/reference/imsegm/superpixels.py:61-63) on crafted label maps, HIP against the oracle AND against the outputs of the real
scikit-image 0.18.3 (``tests/golden/connectivity.npz``), bit for bit.

The 2-D tile path of ``csrc/connectivity.hip`` has hand-over points (more than 256 local components in a tile, a BFS
frontier of more than 64 cells, bounding boxes beyond the LDS tiles, oversize components); every case below is built to
cross one of them.
"""
import importlib.util
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def hip():
    from pyimsegm_amd import _hip
    _hip.default_context()
    return _hip


HERE = os.path.dirname(os.path.abspath(__file__))
_spec = importlib.util.spec_from_file_location('make_golden_connectivity', os.path.join(HERE, 'golden', 'make_golden_connectivity.py'))
GEN = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(GEN)
_salted = GEN.salted

#: the crafted maps of tests/golden/make_golden_connectivity.py (outputs of the real scikit-image 0.18.3 in connectivity.npz)
#: plus one that only the oracle covers
CASES = [(name, (lambda name=name: GEN.make(name)), GEN.CASES[name][1], GEN.CASES[name][2])
         for name in GEN.CASES if name != 'volume'] + [
    ('tiny', lambda: GEN.blocks(4, 4, 2, 2), 2, 100),
]

GOLDEN = np.load(os.path.join(HERE, 'golden', 'connectivity.npz'))

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
        key = '%s_start%d' % (name, start_label)
        if key in GOLDEN:                              # ... and with the real scikit-image 0.18.3
            assert np.array_equal(got, GOLDEN[key])
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


def test_enforce_connectivity_volume_golden(hip):
    """the volume of the golden set (real scikit-image 0.18.3), both start labels"""
    _, min_size, max_size = GEN.CASES['volume']
    for start_label in (0, 1):
        lab = GEN.make('volume') + start_label
        vol = hip.Volume3D(*lab.shape)
        try:
            got = vol.enforce_connectivity(lab, min_size, max_size, start_label)
        finally:
            vol.close()
        assert np.array_equal(got, GOLDEN['volume_start%d' % start_label])


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
