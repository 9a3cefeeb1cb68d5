"""The centroid update INSIDE the assignment kernel (the workgroup that completes a centroid divides its sums; default where the
whole assignment grid is resident at once, IMSEGM_FUSE_FINALIZE for any image) against separate finalize launches and the oracle,
and its hand-back: a tile without a candidate list raises the failure word and the host redoes the sweeps with separate launches.
(Until round 6 this file also held the tests of the one-launch persistent sweeps of round 3, which were removed.)"""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def hip():
    from pyimsegm_amd import _hip
    _hip.load_library()
    return _hip


def _labels(image, sp_size, regul):
    from pyimsegm_amd.superpixels import segment_slic_img2d
    return np.asarray(segment_slic_img2d(image, sp_size, regul))


def test_noise_image_with_wandering_centroids(hip):
    """pure noise: the centroids move far and the windows overlap heavily -- whatever path finishes the image, the label map is
    the oracle's"""
    from oracle import oracle as orc
    image = np.random.default_rng(17).integers(0, 256, (384, 448, 3)).astype(np.uint8)
    assert np.array_equal(_labels(image, 32, 0.05), orc.segment_slic_img2d(image, 32, 0.05))


@pytest.mark.parametrize('shape,sp_size,regul,seed', [((647, 1024), 35, 0.2, 100), ((1030, 1200), 46, 0.2, 5), ((300, 1000), 24, 0.3, 4)])
def test_centroid_update_inside_the_assignment_kernel(hip, monkeypatch, shape, sp_size, regul, seed):
    """per-sweep launches with the centroid update done by the workgroup that completes a centroid (default for small images,
    IMSEGM_FUSE_FINALIZE for any) against separate finalize launches and the oracle"""
    from oracle import oracle as orc
    from pyimsegm_amd.utilities.synthetic import voronoi_image
    image = voronoi_image(shape[0], shape[1], seed=seed)
    monkeypatch.setenv('IMSEGM_FUSE_FINALIZE', '1')
    p0, f0 = hip.slic_sweep_runs()
    fused = _labels(image, sp_size, regul)
    assert hip.slic_sweep_runs()[1] == f0, 'the image was handed back to the separate finalize launches'
    monkeypatch.delenv('IMSEGM_FUSE_FINALIZE')
    monkeypatch.setenv('IMSEGM_SEPARATE_FINALIZE', '1')
    separate = _labels(image, sp_size, regul)
    assert np.array_equal(fused, separate)
    assert np.array_equal(fused, orc.segment_slic_img2d(image, sp_size, regul))


def test_fused_centroid_update_hands_the_image_back(hip, monkeypatch):
    """the hand-back of the centroid update inside the assignment kernel.  Superpixels of 5 pixels put far more than
    SLIC_MAXC = 64 centroids within reach of a 64 x 32 tile: such a tile has no candidate list, its pixels go to the global sums
    past the arrival counts, the kernel raises the failure word and the host redoes the sweeps with separate finalize launches --
    the result must be the oracle's (and that of a run with separate launches from the start), and the hand-back must be counted"""
    from oracle import oracle as orc
    from pyimsegm_amd.utilities.synthetic import voronoi_image
    image = voronoi_image(96, 128, seed=17)
    monkeypatch.setenv('IMSEGM_FUSE_FINALIZE', '1')
    before = hip.slic_sweep_runs()[1]
    fused = _labels(image, 5, 0.2)
    assert hip.slic_sweep_runs()[1] == before + 1, 'the overflowing tiles were expected to hand the image back'
    monkeypatch.delenv('IMSEGM_FUSE_FINALIZE')
    monkeypatch.setenv('IMSEGM_SEPARATE_FINALIZE', '1')
    separate = _labels(image, 5, 0.2)
    assert hip.slic_sweep_runs()[1] == before + 1
    assert np.array_equal(fused, separate)
    assert np.array_equal(fused, orc.segment_slic_img2d(image, 5, 0.2))
