"""The feature table of several descriptor groups formed and KEPT on the device (imsegm_image2d_features_place, the colour
statistics and the Leung-Malik statistics side by side) and the class model evaluated on it with more than 64 features --
against the path that brings every group to the host (compute_selected_features_color2d as the reference concatenates them,
/root/reference/imsegm/descriptors.py:1207-1270) and scikit-learn's predict_proba."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu


def _image(height, width, seed):
    from pyimsegm_amd.utilities.synthetic import voronoi_image
    return voronoi_image(height, width, seed=seed)


@pytest.mark.parametrize('feats', [
    {'color': ('mean', 'std'), 'tLM_short': ('mean', 'std', 'energy')},
    {'tLM_short': ('energy', 'mean')},
    {'tLM_short': ('std', ), 'color': ('energy', )},          # (columns: the colour group first, whatever the order of the keys)
])
def test_resident_table_equals_the_groups_brought_to_the_host(feats):
    from pyimsegm_amd import pipelines as pipe
    from pyimsegm_amd.descriptors import compute_selected_features_img2d
    image = _image(150, 200, seed=5)
    res = pipe._ResidentImage(image, feats, 14, 0.2, features_to_host=False)
    try:
        assert res.resident_features
        table = res.features
        slic = res.slic
    finally:
        res.close()
    expected, names = compute_selected_features_img2d(image, slic, feats)
    assert table.shape == expected.shape == (int(slic.max()) + 1, len(names))
    assert np.array_equal(table, expected)


def test_float_image_and_groups_the_device_does_not_keep():
    from pyimsegm_amd import pipelines as pipe
    from pyimsegm_amd.descriptors import compute_selected_features_img2d, resident_feature_groups
    assert resident_feature_groups({'color': ('mean', 'median')}) is None
    assert resident_feature_groups({'color_hsv': ('mean', )}) is None
    assert resident_feature_groups({'tLM': ('mean', ), 'unknown': ('mean', )}) is None
    assert [g[3] for g in resident_feature_groups({'tLM': ('mean', 'std', 'energy'), 'color': ('mean', )})] == [3, 180]
    image = _image(96, 128, seed=8).astype(np.float64) / 255.
    feats = {'color': ('mean', ), 'tLM_short': ('mean', )}
    res = pipe._ResidentImage(image, feats, 12, 0.3)
    try:
        assert res.resident_features
        expected, _ = compute_selected_features_img2d(image, res.slic, feats)
        assert np.array_equal(res.features, expected)
    finally:
        res.close()


@pytest.mark.parametrize('feats,columns', [({'tLM_short': ('mean', 'std')}, 90), ({'color': ('mean', 'std', 'energy'), 'tLM_short': ('mean', 'std', 'energy')}, 144),
                                           ({'tLM': ('mean', 'std', 'energy'), 'color': ('mean', 'std', 'energy')}, 189)])
def test_mixture_with_many_features_on_the_device(feats, columns):
    """scaler + full-covariance mixture over 90 / 144 / 189 features: probabilities as scikit-learn's, segmentation equal to the
    one under scikit-learn's probabilities"""
    from pyimsegm_amd import _hip
    from pyimsegm_amd import pipelines as pipe
    from pyimsegm_amd.graph_cuts import compute_pairwise_cost, estim_class_model, predict_proba
    image = _image(400, 480, seed=11)
    res = pipe._ResidentImage(image, feats, 10, 0.2)
    try:
        features = res.features
        assert features.shape[1] == columns and features.shape[0] > 4 * columns
        model = estim_class_model(features, 3, 'GMM', None, True)
        gmm = _hip.DeviceGmm(model)
        assert gmm.n_features == columns
        proba = predict_proba(model, features)
        out = res.sess.segment(compute_pairwise_cost(2., (0, 3)), 'model', gmm=gmm, want_proba=True, want_graph_labels=True)
        assert np.allclose(out['proba'], proba, rtol=0, atol=1e-7), float(np.abs(out['proba'] - proba).max())
        on_device, _ = res.segment(None, 2., 'model', model=model, want_soft=False)
        on_host, _ = res.segment(proba, 2., 'model', want_soft=False)
        assert np.array_equal(on_device, on_host)
        assert len(np.unique(on_device)) > 1
    finally:
        res.close()


def test_placement_is_checked():
    from pyimsegm_amd import _hip
    image = _image(64, 80, seed=2)
    sess = _hip.Image2D(64, 80).upload(image)
    try:
        sess.slic(30, 10.)
        with pytest.raises(_hip.HipError):
            sess.features_place(9, 9)
        with pytest.raises(_hip.HipError):
            sess.get_features(9)                         # no table yet
        sess.features_place(12, 6)
        with pytest.raises(_hip.HipError):
            sess.features_color(True, True, True, to_host=False)          # 9 columns at column 6 of 12
        sess.features_place(12, 3)
        with pytest.raises(_hip.HipError):
            sess.features_color(True, False, False, to_host=True)         # a placed group stays on the device
        first = sess.features_color(True, True, False, to_host=True)      # (the placement was consumed by the failed call)
        sess.features_place(9, 0).features_color(True, True, False, to_host=False)
        sess.features_place(9, 6).features_color(False, False, True, to_host=False)
        table = sess.get_features(9)
        assert np.array_equal(table[:, :6], first)
        assert np.array_equal(table[:, 6:], sess.features_color(False, False, True, to_host=True))
        with pytest.raises(_hip.HipError):
            sess.get_features(9)                         # the table is the 3-column one now
    finally:
        sess.close()


def test_texture_feature_sets_with_images_in_flight():
    """the resident table (colour + Leung-Malik groups) built by worker threads -- one context, one HIP stream and one recycled
    session each, the factorised bank shared between them -- gives the features and segmentations of one image at a time"""
    from pyimsegm_amd import pipelines as pipe
    images = [_image(150, 190, seed=60 + i) for i in range(5)]
    feats = {'color': ('mean', ), 'tLM_short': ('mean', 'std')}
    np.random.seed(0)
    model, fts_seq = pipe.estim_model_classes_group(images, 3, feats, sp_size=14, sp_regul=0.2, nb_workers=1)
    np.random.seed(0)
    _, fts_par = pipe.estim_model_classes_group(images, 3, feats, sp_size=14, sp_regul=0.2, nb_workers=3)
    assert fts_seq[0].shape[1] == 3 + 90
    assert all(np.array_equal(a, b) for a, b in zip(fts_seq, fts_par))
    seq = [pipe.segment_color2d_slic_features_model_graphcut(im, model, feats, sp_size=14, sp_regul=0.2, gc_regul=1.5)[0] for im in images]
    par = pipe.segment_batch_color2d_slic_features_model_graphcut(images, model, feats, sp_size=14, sp_regul=0.2, gc_regul=1.5, nb_workers=3)
    assert len(par) == len(seq) and all(np.array_equal(a, b) for a, b in zip(seq, par))
    assert any(len(np.unique(s)) > 1 for s in seq)
