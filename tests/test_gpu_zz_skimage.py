"""The HIP path (public API of pyimsegm_amd.superpixels) against outputs of the REAL scikit-image 0.18.3
(tests/golden/skimage.npz; see tests/test_golden_skimage.py for the oracle side)."""
import numpy as np
import pytest

from test_golden_skimage import GEN, VEC, make_input

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('name', sorted(GEN.CASES_2D))
def test_hip_slic2d_equals_scikit_image(name):
    from pyimsegm_amd import superpixels as sp_mod
    expr, sp, rc = GEN.CASES_2D[name]
    img = make_input(name, expr)
    labels = sp_mod.segment_slic_img2d(img, sp, rc)
    assert labels.dtype == np.int64 and np.array_equal(labels, VEC[name + '_final'])
    if name + '_slico_final' in VEC.files:
        assert np.array_equal(sp_mod.segment_slic_img2d(img, sp, rc, slico=True), VEC[name + '_slico_final'])
    if name + '_centroids' in VEC.files:
        centres = np.array(sp_mod.superpixel_centers(labels), dtype=np.float64)
        np.testing.assert_allclose(centres, VEC[name + '_centroids'], rtol=0, atol=1e-12)


@pytest.mark.parametrize('name', sorted(GEN.CASES_3D))
def test_hip_slic3d_equals_scikit_image(name):
    from pyimsegm_amd import superpixels as sp_mod
    expr, sp, rc, space = GEN.CASES_3D[name]
    vol = make_input(name, expr)
    assert np.array_equal(sp_mod.segment_slic_img3d_gray(vol, sp, rc, space), VEC[name + '_label'])


@pytest.mark.parametrize('name', sorted(GEN.CASES_3D_F32))
def test_hip_float32_volume_equals_scikit_image(name):
    """a float32 volume stays float32 on the device as inside scikit-image 0.18: float32 filter output, distances and the
    raster-order float32 running sums of the centroid update (one wave per centroid, csrc/volume.hip)"""
    from pyimsegm_amd import _hip
    from pyimsegm_amd import superpixels as sp_mod
    expr, sp, rc, space = GEN.CASES_3D_F32[name]
    vol = make_input(name, expr)
    assert vol.dtype == np.float32
    # raw SLIC (before measure.label) and the relabelled map of superpixels.py:111
    n_seg, compact = sp_mod._slic3d_params(vol.shape, sp, rc, space)
    sess = _hip.Volume3D(*vol.shape).upload(vol)
    sess.slic(n_seg, compact, sigma=1., spacing=space, start_label=0)
    raw = sess.get_labels()
    sess.close()
    assert np.array_equal(raw, VEC[name + '_slic'])
    assert np.array_equal(sp_mod.segment_slic_img3d_gray(vol, sp, rc, space), VEC[name + '_label'])
