"""HIP path (through the C ABI) against golden vectors produced by the reference's own code
(tests/golden/make_golden.py)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def _load(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def test_graph_and_centres_match_reference():
    from pyimsegm_amd import superpixels as sp
    g = _load('graph.npz')
    v, e = sp.make_graph_segm_connect_grid2d_conn4(g['seg2d'])
    assert v.tolist() == g['seg2d_vertices'].tolist() and e == g['seg2d_edges'].tolist()
    v, e = sp.make_graph_segm_connect_grid3d_conn6(g['seg3d'])
    assert v.tolist() == g['seg3d_vertices'].tolist() and e == g['seg3d_edges'].tolist()
    centres = np.array(sp.superpixel_centers(g['seg3d']), dtype=np.float64)
    np.testing.assert_allclose(centres, g['seg3d_centres'], rtol=0, atol=1e-12)


def test_descriptors_match_reference_cython():
    from pyimsegm_amd import descriptors as d
    t = _load('descriptors.npz')
    np.testing.assert_allclose(d.cython_img2d_color_mean(t['img2d'], t['seg2d']), t['mean2d'], rtol=1e-6, atol=1e-5)
    np.testing.assert_allclose(d.cython_img2d_color_energy(t['img2d'], t['seg2d']), t['energy2d'], rtol=1e-6, atol=1e-4)
    np.testing.assert_allclose(d.cython_img2d_color_std(t['img2d'], t['seg2d'])**2, t['var2d'], rtol=1e-6, atol=1e-4)
    np.testing.assert_allclose(d.cython_img3d_gray_mean(t['vol'], t['segv']), t['meanv'], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(d.cython_img3d_gray_energy(t['vol'], t['segv']), t['energyv'], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(d.cython_img3d_gray_std(t['vol'], t['segv'])**2, t['varv'], rtol=1e-6, atol=1e-7)
