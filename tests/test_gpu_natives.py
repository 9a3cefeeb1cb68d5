"""HIP versions of the remaining natives (batched label histograms, batched ray features, cut_grid_graph) against the
reference's own compiled features_cython.pyx (oracle/_ref), bit for bit, and against the reference's doctest vectors."""
import numpy as np
import pytest

import natives_cases as NC

pytestmark = pytest.mark.gpu


def test_label_hist_doctests_and_random_windows(ref_cython):
    from pyimsegm_amd import descriptors as D
    segm = NC.hist_segmentation()
    assert D.cython_label_hist_seg2d(segm[2:5, 4:7], np.ones((3, 3)), 3).tolist() == [0., 5., 4.]
    assert D.cython_label_hist_seg2d(segm[1:6, 3:8], np.ones((5, 5)), 3).tolist() == [0., 19., 6.]
    hist, size = D.compute_label_hist_segm(segm, [6, 6], np.ones((3, 3)), 3)
    assert hist.tolist() == [0., 7., 2.] and size == 9.0                       # descriptors.py:1436-1439
    hist, size = D.compute_label_hist_segm(segm, [4, 4], np.ones((5, 5)), 3)
    assert hist.tolist() == [0., 17., 8.] and size == 25.0
    rng = np.random.default_rng(5)
    big = rng.integers(-1, 7, (90, 130)).astype(np.int16)
    rr, cc = np.mgrid[-6:7, -6:7]
    selem = ((rr**2 + cc**2) <= 36).astype(np.int16)
    positions = np.stack([rng.integers(0, 90, 40), rng.integers(0, 130, 40)], axis=1)
    hists, sizes = D.compute_label_hist_positions(big, positions, selem, 7)
    for pos, hist, size in zip(positions, hists, sizes):
        b0, e0, s0, s1 = D.adjust_bounding_box_crop(big.shape, selem.shape, pos)
        sel, se = big[b0[0]:e0[0], b0[1]:e0[1]], selem[s0[0]:s1[0], s0[1]:s1[1]]
        ref = [np.sum(np.logical_and(sel == lb, se == 1)) for lb in range(7)]
        assert hist.tolist() == ref and size == se.sum()
        if ref_cython is not None:
            cy = np.array(ref_cython.computeLabelHistogram2d(np.ascontiguousarray(sel), np.ascontiguousarray(se), 7))
            assert hist.tolist() == cy.tolist()


def test_ray_features_bit_exact_and_doctests(ref_cython):
    from pyimsegm_amd import descriptors as D
    seg = NC.disc_segmentation()
    assert D.cython_ray_features_seg2d(np.zeros((100, 150), dtype=bool), (50, 75), 90).tolist() == [-1.] * 4
    for position, step, expect in NC.RAY_DOCTESTS:
        assert D.cython_ray_features_seg2d(seg, position, step).astype(int).tolist() == expect
    assert np.round(D.compute_ray_features_segm_2d(seg, (60, 40), 30, smooth_coef=1)).tolist() == \
        [66.0, 52.0, 32.0, 16.0, 8.0, 5.0, 5.0, 8.0, 16.0, 33.0, 53.0, 67.0]    # descriptors.py:1740-1741
    if ref_cython is None:
        pytest.skip('oracle/_ref not built')
    rng = np.random.default_rng(9)
    blobs = rng.random((120, 170)) > 0.995
    from scipy import ndimage
    blobs = ndimage.binary_dilation(blobs, iterations=6)
    positions = np.stack([rng.integers(0, 120, 60), rng.integers(0, 170, 60)], axis=1)
    for edge in ('up', 'down'):
        for step in (5., 12.5, 30.):
            got = D.hip_ray_features_positions(blobs, positions, step, edge)
            for pos, rays in zip(positions, got):
                ref = np.array(ref_cython.computeRayFeaturesBinary2d(np.array(blobs, dtype=np.int8), np.array(pos, dtype=np.int32),
                                                                     float(step), {'up': 1, 'down': -1}[edge]))
                assert np.array_equal(rays, ref), (edge, step, pos.tolist())


def test_cut_grid_graph_known_answers_and_oracle(oracle):
    from pyimsegm_amd import graph_cuts as G
    import gco
    for (gc_regul, seed, coef), expect in (((0., 0, 0.5), NC.GRID_EXPECT_SHAPE), ((.5, 1, 0.), NC.GRID_EXPECT_SEED)):
        unary, pairwise, cost_v, cost_h = NC.grid_problem(gc_regul, seed, coef)
        labels = gco.cut_grid_graph(unary, pairwise, cost_v, cost_h, n_iter=999)
        assert labels.dtype == np.int32 and np.array_equal(labels.reshape(NC.GRID_SEGM.shape), expect)
    rng = np.random.default_rng(4)
    unary = rng.random((23, 31, 4))
    pairwise = (1 - np.eye(4)) * 0.4
    cost_v, cost_h = rng.random((22, 31)) + 0.2, rng.random((23, 30)) + 0.2
    assert np.array_equal(G.cut_grid_graph(unary, pairwise, cost_v, cost_h), oracle.cut_grid_graph(unary, pairwise, cost_v, cost_h))


@pytest.mark.parametrize('dtype', [np.uint8, np.float32, np.float64])
def test_device_median_and_mean_gradient_2d(dtype):
    """'median' / 'meanGrad' of compute_image2d_color_statistic on the device against the numpy route of the reference
    (np.median of the per-label lists; np.gradient sums stored in the image's dtype, then the float32-staged mean)"""
    from pyimsegm_amd import _hip
    from pyimsegm_amd import descriptors as D
    rng = np.random.default_rng(3)
    img = rng.random((67, 93, 3)) * 255
    img = img.astype(dtype) if dtype == np.uint8 else (img / 255.).astype(dtype)
    seg = (np.arange(67)[:, None] // 9) * 11 + (np.arange(93)[None, :] // 9)
    seg[seg == 5] = 4                                        # label 5 has no pixel
    seg[10:14, 20:23] = 80                                   # an even and an odd sized island
    sess = _hip.Image2D(67, 93).upload(img).set_labels(seg)
    med, grad = sess.median(), sess.mean_gradient()
    sess.close()
    # the reference's loop (descriptors.py:420-455) restated with numpy: per label and channel np.median of the values
    ref = np.full((seg.max() + 1, 3), np.nan)
    for k in np.unique(seg):
        ref[k] = [np.median(img[:, :, c][seg == k]) for c in range(3)]
    assert np.array_equal(np.isnan(med), np.isnan(ref)) and np.array_equal(np.nan_to_num(med), np.nan_to_num(ref))
    gimg = np.zeros_like(img)
    for c in range(3):
        gimg[:, :, c] = np.sum(np.gradient(img[:, :, c]), axis=0)
    ref_g = D.hip_img2d_color_mean(gimg, seg)
    assert np.array_equal(grad, ref_g)
    feats, names = D.compute_image2d_color_statistic(img, seg, ('mean', 'median', 'meanGrad'))
    assert feats.shape == (seg.max() + 1, 9) and np.array_equal(feats[:, 3:6], np.nan_to_num(ref))
    assert np.array_equal(feats[:, 6:9], np.nan_to_num(ref_g))


@pytest.mark.parametrize('dtype', [np.uint8, np.float64])
def test_device_median_and_mean_gradient_3d(dtype):
    from pyimsegm_amd import descriptors as D
    rng = np.random.default_rng(8)
    vol = rng.random((5, 31, 40))
    vol = (vol * 255).astype(np.uint8) if dtype == np.uint8 else vol
    seg = (np.arange(31)[None, :, None] // 8) * 6 + (np.arange(40)[None, None, :] // 8) + np.arange(5)[:, None, None] // 3 * 30
    feats, names = D.compute_image3d_gray_statistic(vol, seg, ('mean', 'median', 'meanGrad'))
    assert names == ['gray_mean', 'gray_median', 'gray_meanGrad']
    ref_med = np.nan_to_num(np.array([np.median(vol[seg == k]) if np.any(seg == k) else np.nan for k in range(seg.max() + 1)]))
    assert np.array_equal(feats[:, 1], ref_med)
    grad = np.zeros_like(vol)
    for i in range(vol.shape[0]):
        grad[i] = np.sum(np.gradient(vol[i]), axis=0)
    assert np.array_equal(feats[:, 2], D.cython_img3d_gray_mean(grad, seg))


def test_reference_statistic_doctests_with_all_flags():
    """descriptors.py:714-736 (3-D, five flags incl. empty labels) and :796-813 (2-D, 15 columns)"""
    from pyimsegm_amd import descriptors as D
    image = np.zeros((2, 3, 8))
    image[0, :, 2:6] = 1
    image[1, :, 3:7] = 3
    segm = np.array([[[0, 0, 0, 0, 1, 1, 1, 1]] * 3, [[2, 2, 2, 2, 5, 5, 5, 5]] * 3])
    features, names = D.compute_image3d_gray_statistic(image, segm)
    assert np.round(features, 3).tolist() == [[0.5, 0.5, 0.5, 0.5, 0.25], [0.5, 0.5, 0.5, 0.5, -0.25], [0.75, 1.299, 2.25, 0.0, 0.75],
                                              [0.0, 0.0, 0.0, 0.0, 0.0], [0.0, 0.0, 0.0, 0.0, 0.0], [2.25, 1.299, 6.75, 3.0, -1.125]]
    image = np.zeros((2, 10, 3))
    image[:, 2:6, 0] = 1
    image[:, 3:7, 1] = 3
    image[:, 4:9, 2] = 2
    segm = np.array([[0, 0, 0, 0, 0, 1, 1, 1, 1, 1], [0, 0, 0, 0, 0, 1, 1, 1, 1, 1]])
    features, names = D.compute_image2d_color_statistic(image, segm)
    assert features.shape == (2, 15) and names[9:12] == ['color-ch1_median', 'color-ch2_median', 'color-ch3_median']
    assert np.array_equal(features[:, 9:12], [[np.median(image[segm == k][:, c]) for c in range(3)] for k in range(2)])
