"""Leung-Malik path at size (BASELINE config 3 works on 2048 x 2048): a 1024 x 1024 image on the device against scipy on a
crop of the same image -- the high-pass on the whole image (its sigma = 150 kernel reaches 600 pixels), the 33 x 33
batteries on interior pixels of a 256 x 256 window -- and the 3-D variant of the texture descriptors against its scipy
formulation (descriptors.py:969-1038)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_lm_responses_at_size_match_scipy_on_a_crop():
    from scipy import ndimage
    from pyimsegm_amd import _hip
    from pyimsegm_amd import descriptors as D
    from pyimsegm_amd.utilities.synthetic import voronoi_image
    size, y0, x0, win, r = 1024, 300, 610, 256, 16
    img = voronoi_image(size, size, seed=5)
    high = img - ndimage.gaussian_filter(img.astype(float), 150)             # descriptors.py:1078 (all three axes)
    sess = _hip.Image2D(size, size).upload(img).set_labels(np.zeros((size, size), dtype=np.int32))
    sess.lm_prepare(150.)
    filters, names = D.create_filter_bank_lm_2d()
    crop = np.rollaxis(high[y0 - r:y0 + win + r, x0 - r:x0 + win + r], -1, 0)    # window + the kernel radius
    for index in (0, 1, 7, 19):                                           # 8-kernel edge / bar batteries, a Gaussian, a LoG
        norm = sess.lm_battery(filters[index], D.MAX_SIGNAL_RESPONSE)
        resp = sess.get_response()
        ref = D.compute_img_filter_response3d(crop, filters[index])[:, r:-r, r:-r]
        got = resp[:, y0:y0 + win, x0:x0 + win]
        scale = max(1.0, np.abs(ref).max())
        assert np.max(np.abs(got - ref)) < 1e-9 * scale, (names[index], np.max(np.abs(got - ref)))
        assert abs(norm - np.sqrt(np.sum(resp**2))) < 1e-9 * norm           # the global L2 norm over 3 x 1024 x 1024 values
    sess.close()


def test_lm_descriptors_of_a_volume_match_scipy():
    from pyimsegm_amd import descriptors as D
    rng = np.random.default_rng(2)
    vol = rng.random((4, 70, 90))
    seg = (np.arange(70)[None, :, None] // 24) * 4 + (np.arange(90)[None, None, :] // 24) + np.arange(4)[:, None, None] // 2 * 12
    flags = ['mean', 'std', 'energy']
    fts, names = D.compute_texture_desc_lm_img3d_val(vol, seg, flags, bank_type='short')
    high = D.image_subtract_gauss_smooth(vol, 150)
    filters, fl_names = D._select_bank('short')
    ref = []
    for battery, fl_name in zip(filters, fl_names):
        resp = D._normalise_response(D.compute_img_filter_response3d(high, battery))
        ref.append(D.compute_image3d_gray_statistic(resp, seg, flags, fl_name)[0])
    ref = np.concatenate(ref, axis=1)
    assert fts.shape == ref.shape == (seg.max() + 1, 15 * 3)
    assert names[0] == 'tLM_sigma1.4-edge_mean' and names[-1] == 'tLM_sigma4.0-GaussLap2_energy'
    assert np.max(np.abs(fts - ref)) < 1e-6 * max(1.0, np.abs(ref).max()), np.max(np.abs(fts - ref))
