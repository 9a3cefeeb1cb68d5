"""GPU tests of the host-side mirror of the reference API (pyimsegm_amd.superpixels / descriptors /
graph_cuts / pipelines): the reference's own doctest vectors, evaluated through the HIP path."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _doctest_image():
    image = np.zeros((2, 10, 3))
    image[:, 2:6, 0] = 1
    image[:, 3:7, 1] = 3
    image[:, 4:9, 2] = 2
    segm = np.array([[0, 0, 0, 0, 0, 1, 1, 1, 1, 1], [0, 0, 0, 0, 0, 1, 1, 1, 1, 1]])
    return image, segm


def test_descriptor_doctests():
    from pyimsegm_amd import descriptors as D
    image, segm = _doctest_image()
    # descriptors.py:225-226, 253-254, 282-283
    assert np.array_equal(D.cython_img2d_color_mean(image, segm), [[0.6, 1.2, 0.4], [0.2, 1.2, 1.6]])
    assert np.array_equal(D.cython_img2d_color_energy(image, segm), [[0.6, 3.6, 0.8], [0.2, 3.6, 3.2]])
    std = D.cython_img2d_color_std(image, segm)
    assert np.allclose(std, [[0.48989794, 1.46969383, 0.80000003], [0.40000001, 1.46969383, 0.80000001]], atol=5e-9, rtol=0)
    # descriptors.py:796-813 (15 columns)
    features, names = D.compute_image2d_color_statistic(image, segm)
    assert names[:3] == ['color-ch1_mean', 'color-ch2_mean', 'color-ch3_mean'] and len(names) == 15
    expect = [[0.6, 1.2, 0.4, 0.5, 1.5, 0.8, 0.6, 3.6, 0.8, 1.0, 0.0, 0.0, 0.2, 0.6, 0.4],
              [0.2, 1.2, 1.6, 0.4, 1.5, 0.8, 0.2, 3.6, 3.2, 0.0, 0.0, 2.0, -0.2, -0.6, -0.6]]
    assert np.round(features, 1).tolist() == expect
    # descriptors.py:1215-1239
    fts, _ = D.compute_selected_features_color2d(image, segm, {'color': ('mean', 'std', 'median')})
    assert np.allclose(np.round(fts, 3), [[0.6, 1.2, 0.4, 0.49, 1.47, 0.8, 1., 0., 0.], [0.2, 1.2, 1.6, 0.4, 1.47, 0.8, 0., 0., 2.]])
    fts, _ = D.compute_selected_features_color2d(image, segm, {'color_hsv': ('mean', 'std')})
    assert np.allclose(np.round(fts, 3), [[0.139, 0.533, 1.4, 0.176, 0.452, 1.356], [0.439, 0.733, 2., 0.244, 0.389, 1.095]])
    fts, _ = D.compute_selected_features_color2d(image, segm, {'tLM_short': ('mean', 'energy')})
    assert fts.shape == (2, 90)
    with pytest.raises(TypeError):
        D.cython_img2d_color_mean(np.zeros((125, 150, 3)), np.zeros((150, 125), dtype=int))


def test_texture_doctest_shapes():
    from pyimsegm_amd import descriptors as D
    h, w, step = 30, 20, 5
    np.random.seed(0)
    seg = (np.arange(h)[:, None] // step) * (w // step) + np.arange(w)[None, :] // step
    img = np.random.random((h, w, 3))
    features, names = D.compute_texture_desc_lm_img2d_clr(img, seg, ['mean', 'std', 'median'], bank_type='short')
    assert features.shape == (24, 135)                         # descriptors.py:1063-1064
    assert names[0] == 'tLM_sigma1.4-edge-ch1_mean' and names[-1] == 'tLM_sigma4.0-GaussLap2-ch3_median'


def test_superpixels_api(oracle):
    from pyimsegm_amd import superpixels as S
    np.random.seed(0)
    img = np.random.random((100, 150, 3))
    slic = S.segment_slic_img2d(img, 20, 0.2)                   # superpixels.py:32-36
    assert slic.shape == (100, 150) and slic.dtype == np.int64
    assert np.array_equal(slic, oracle.segment_slic_img2d(img, 20, 0.2))
    gray = np.random.random((150, 100))
    slic = S.segment_slic_img2d(gray, 20, 0.2)                  # superpixels.py:37-40
    assert slic.shape == (150, 100)
    assert np.array_equal(slic, oracle.segment_slic_img2d(gray, 20, 0.2))
    grid = np.array([[0] * 5 + [1] * 5, [2] * 5 + [3] * 5])
    v, edges = S.make_graph_segm_connect_grid2d_conn4(grid)     # superpixels.py:163-168
    assert v.tolist() == [0, 1, 2, 3] and edges == [[0, 1], [0, 2], [1, 3], [2, 3]]
    segm = np.array([[0] * 6 + [1] * 5, [0] * 6 + [2] * 5])
    assert S.superpixel_centers(segm) == [(0.5, 2.5), (0.0, 8.0), (1.0, 8.0)]     # superpixels.py:211-213
    assert S.superpixel_centers(np.array([segm, segm, segm])) == [[1.0, 0.5, 2.5], [1.0, 0.0, 8.0], [1.0, 1.0, 8.0]]


def test_edge_weight_doctests():
    """graph_cuts.py:587-609: every edge type on seeded inputs"""
    from pyimsegm_amd import graph_cuts as G
    segments = np.array([[0] * 3 + [1] * 5 + [2] * 4, [4] * 4 + [5] * 5 + [6] * 3])
    np.random.seed(0)
    img = np.random.random(segments.shape + (3, )) * 255
    features = np.random.random((segments.max() + 1, 15)) * 10
    proba = np.random.random((segments.max() + 1, 2))
    edges, weights = G.compute_edge_weights(segments)
    assert edges.dtype == np.int32
    assert edges.tolist() == [[0, 1], [1, 2], [0, 4], [1, 4], [1, 5], [2, 5], [4, 5], [2, 6], [5, 6]]
    assert np.round(weights, 2).tolist() == [1.0] * 9
    _, weights = G.compute_edge_weights(segments, image=img, edge_type='spatial')
    assert np.round(weights, 3).tolist() == [0.776, 0.69, 2.776, 0.853, 2.194, 0.853, 0.69, 2.776, 0.776]
    _, weights = G.compute_edge_weights(segments, image=img, edge_type='color')
    assert np.round(weights, 3).tolist() == [0.06, 0.002, 0.001, 0.001, 0.001, 0.009, 0.001, 0.019, 0.044]
    _, weights = G.compute_edge_weights(segments, features=features, edge_type='features')
    assert np.round(weights, 3).tolist() == [0.031, 0.005, 0.051, 0.032, 0.096, 0.013, 0.018, 0.033, 0.013]
    _, weights = G.compute_edge_weights(segments, proba=proba, edge_type='model')
    assert np.round(weights, 3).tolist() == [0.001, 0.028, 1.122, 0.038, 0.117, 0.688, 0.487, 1.152, 0.282]
    with pytest.raises(ValueError):
        G.compute_edge_weights(segments, edge_type='model')
    with pytest.raises(RuntimeError):
        G.compute_edge_weights(segments, edge_type='color')


def test_segment_graph_cut_general_doctests():
    """graph_cuts.py:680-713"""
    from pyimsegm_amd import graph_cuts as G
    np.random.seed(0)
    segments = np.array([[0] * 3 + [2] * 3 + [4] * 3 + [6] * 3 + [8] * 3, [1] * 3 + [3] * 3 + [5] * 3 + [7] * 3 + [9] * 3])
    proba = np.array([[0.1] * 6 + [0.9] * 4, [0.9] * 6 + [0.1] * 4], dtype=float).T
    proba += (0.5 - np.random.random(proba.shape)) * 0.2
    out = G.segment_graph_cut_general(segments, proba, gc_regul=0., edge_type='')
    assert out.tolist() == [1, 1, 1, 1, 1, 1, 0, 0, 0, 0]
    labels = G.segment_graph_cut_general(segments, proba, gc_regul=1., edge_type='spatial')
    assert labels.dtype == np.int32
    assert np.array_equal(labels[segments], [[1] * 9 + [0] * 6] * 2)
    dbg = {}
    G.segment_graph_cut_general(segments, proba, gc_regul=1., edge_type='model', debug_visual=dbg)
    assert {'segments', 'edges', 'edge_weights'} <= set(dbg)
    transitions = G.count_label_transitions_connected_segments(
        {'a': np.array([[0] * 3 + [1] * 3 + [2] * 3 + [3] * 3 + [4] * 3, [5] * 3 + [6] * 3 + [7] * 3 + [8] * 3 + [9] * 3])},
        {'a': np.array([0, 0, 1, 1, 2, 0, 1, 1, 0, 2])})
    assert transitions.tolist() == [[2., 5., 1.], [5., 3., 1.], [1., 1., 1.]]     # graph_cuts.py:768-771


def _oracle_pipeline(oracle, image, model, sp_size, sp_regul, gc_regul):
    from pyimsegm_amd import graph_cuts as G
    slic = oracle.segment_slic_img2d(image, sp_size, sp_regul)
    img32, seg32 = np.asarray(image, dtype=np.float32), slic.astype(np.int32)
    mean = oracle.color2d_mean(img32, seg32)
    std = np.sqrt(oracle.color2d_variance(img32, seg32, mean.astype(np.float32)))
    energy = oracle.color2d_energy(img32, seg32)
    features = np.nan_to_num(np.hstack([mean, std, energy]))
    proba = model.predict_proba(features)
    _, edges = oracle.adjacency(seg32)
    edges = np.array(edges, dtype=np.int32)
    weights = G.compute_edge_model(edges, proba, 'lT')
    weights = weights / G.compute_spatial_dist([tuple(c) for c in oracle.centers(seg32)], edges, relative=True)
    weights = np.clip(weights, 1e-3, 1e3)
    labels = oracle.cut_general_graph(edges, weights, G.compute_unary_cost(proba), G.compute_pairwise_cost(gc_regul, proba.shape))
    return slic, features, labels[slic], proba[slic]


@pytest.mark.parametrize('size,sp', [(256, 18), (600, 25)])
def test_pipeline_equals_oracle_pipeline(oracle, size, sp):
    """end to end: same model -> identical segmentation and soft segmentation as the CPU oracle path"""
    from pyimsegm_amd import pipelines as P
    from pyimsegm_amd.descriptors import FEATURES_SET_COLOR
    from pyimsegm_amd.utilities.synthetic import voronoi_image
    image = voronoi_image(size, size + 40, seed=7)
    np.random.seed(0)
    model, list_features = P.estim_model_classes_group([image], 3, FEATURES_SET_COLOR, sp_size=sp, sp_regul=0.2)
    segm, soft = P.segment_color2d_slic_features_model_graphcut(image, model, FEATURES_SET_COLOR, sp_size=sp,
                                                                sp_regul=0.2, gc_regul=2.0, gc_edge_type='model')
    slic_ref, fts_ref, segm_ref, soft_ref = _oracle_pipeline(oracle, image, model, sp, 0.2, 2.0)
    assert np.allclose(list_features[0], fts_ref, rtol=1e-10, atol=1e-12)
    assert np.array_equal(segm, segm_ref)
    assert np.allclose(soft, soft_ref, rtol=1e-9, atol=1e-12)
    assert soft.shape == image.shape[:2] + (3, )
    # doctest of pipelines.py:76-83
    np.random.seed(0)
    img = np.random.random((125, 150, 3)) / 2.
    img[:, :75] += 0.5
    segm, seg_soft = P.pipe_color2d_slic_features_model_graphcut(img, 2, {'color': ['mean']})
    assert segm.shape == (125, 150) and seg_soft.shape == (125, 150, 2)
    assert len(np.unique(segm[:, :60])) == 1 and len(np.unique(segm[:, 90:])) == 1 and segm[0, 0] != segm[0, -1]
    with pytest.raises(ValueError):
        P.compute_color2d_superpixels_features(img, {'color': ['mean']}, sp_regul=0.)


def _lm_host_reference(img, seg, flags, bank_type):
    """the reference's scipy formulation (descriptors.py:1041-1106) with numpy float32-staged statistics"""
    from scipy import ndimage
    from pyimsegm_amd import descriptors as D
    high = img - ndimage.gaussian_filter(img.astype(float), 150)
    roll = np.rollaxis(high, -1, 0)
    filters, names = D._select_bank(bank_type)
    out = []
    nb = seg.max() + 1
    cnt = np.bincount(seg.ravel(), minlength=nb).astype(float)
    for battery in filters:
        resp = D._normalise_response(D.compute_img_filter_response3d(roll, battery))
        r32 = np.rollaxis(resp, 0, 3).astype(np.float32)
        cols = []
        mean = np.stack([np.bincount(seg.ravel(), weights=r32[..., c].ravel().astype(np.float64), minlength=nb) / cnt
                         for c in range(3)], axis=1)
        if 'mean' in flags:
            cols.append(mean)
        if 'std' in flags:
            d = r32 - mean.astype(np.float32)[seg]
            cols.append(np.sqrt(np.stack([np.bincount(seg.ravel(), weights=(d[..., c] * d[..., c]).ravel().astype(np.float64),
                                                      minlength=nb) / cnt for c in range(3)], axis=1)))
        if 'energy' in flags:
            cols.append(np.stack([np.bincount(seg.ravel(), weights=(r32[..., c] * r32[..., c]).ravel().astype(np.float64),
                                              minlength=nb) / cnt for c in range(3)], axis=1))
        out.append(np.hstack(cols))
    return np.concatenate(out, axis=1)


@pytest.mark.parametrize('shape,bank,dtype', [((60, 75), 'short', 'f64'), ((97, 130), 'normal', 'u8'),
                                              # narrower / lower than a kernel (33): the reflected border is crossed more than once
                                              ((12, 200), 'normal', 'u8'), ((130, 17), 'short', 'f64'), ((101, 16), 'normal', 'f64')])
def test_texture_on_device_matches_scipy(shape, bank, dtype):
    """Leung-Malik features computed by the HIP kernels vs the scipy formulation of the reference"""
    from pyimsegm_amd import descriptors as D
    rng = np.random.default_rng(3)
    if dtype == 'u8':
        img = rng.integers(0, 256, shape + (3, )).astype(np.uint8)
    else:
        img = rng.random(shape + (3, ))
    seg = (np.arange(shape[0])[:, None] // 20) * ((shape[1] + 24) // 25) + np.arange(shape[1])[None, :] // 25
    flags = ['mean', 'std', 'energy']
    fts, names = D.compute_texture_desc_lm_img2d_clr(img, seg, flags, bank_type=bank)
    ref = _lm_host_reference(img, seg, flags, bank)
    nb_bat = 15 if bank == 'short' else 20
    assert fts.shape == ref.shape == (seg.max() + 1, nb_bat * 9)
    assert names[0] == 'tLM_sigma1.4-edge-ch1_mean'
    scale = np.abs(ref).max()
    assert np.max(np.abs(fts - ref)) < 1e-5 * max(scale, 1.0), np.max(np.abs(fts - ref))


@pytest.mark.parametrize('flags', [('mean', 'std', 'energy'), ('std', ), ('energy', 'mean')])
def test_texture_one_call_equals_battery_by_battery(flags, monkeypatch):
    """imsegm_image2d_lm_features (all batteries in one call, the L2 norm of a battery stays on the device) against the
    battery-by-battery calls with the norm on the host: same table up to the last bits of log()"""
    from pyimsegm_amd import _hip, descriptors as D
    rng = np.random.default_rng(8)
    img = rng.integers(0, 256, (90, 141, 3)).astype(np.uint8)
    img[:, 70:] //= 3
    seg = ((np.arange(90)[:, None] // 18) * 6 + np.arange(141)[None, :] // 24).astype(np.int32)
    filters, fl_names = D.create_filter_bank_lm_2d()
    sess = _hip.Image2D(90, 141).upload(img).set_labels(seg)
    sess.lm_prepare(150.)
    one = sess.lm_features(filters, D.MAX_SIGNAL_RESPONSE, mean='mean' in flags, std='std' in flags, energy='energy' in flags)
    blocks = []
    for battery in filters:
        norm = sess.lm_battery(battery, D.MAX_SIGNAL_RESPONSE)
        m, e, v = sess.response_stats(np.log(1 + norm) / 0.03, norm)
        blocks += ([m] if 'mean' in flags else []) + ([np.sqrt(v)] if 'std' in flags else []) + ([e] if 'energy' in flags else [])
    sess.close()
    ref = np.hstack(blocks)
    assert one.shape == ref.shape == (seg.max() + 1, 3 * len(flags) * len(filters))
    assert np.max(np.abs(one - ref)) <= 1e-11 * max(1.0, np.abs(ref).max()), np.max(np.abs(one - ref))
    # ... the separable kernels of the bank (28 of 76: two 33-tap passes instead of a 33 x 33 sum) against the all-dense evaluation
    sess = _hip.Image2D(90, 141).upload(img).set_labels(seg)
    sess.lm_prepare(150.)
    dense = sess.lm_features(filters, D.MAX_SIGNAL_RESPONSE, mean='mean' in flags, std='std' in flags, energy='energy' in flags,
                             separable=False)
    # ... and the mirror pairs of an edge / bar battery (two multiply-adds per four pixels and pair) against the kernels one by one
    single = sess.lm_features(filters, D.MAX_SIGNAL_RESPONSE, mean='mean' in flags, std='std' in flags, energy='energy' in flags,
                              mirror=False)
    assert np.max(np.abs(one - dense)) <= 1e-9 * max(1.0, np.abs(dense).max()), np.max(np.abs(one - dense))
    assert np.max(np.abs(one - single)) <= 1e-9 * max(1.0, np.abs(single).max()), np.max(np.abs(one - single))
    # ... the separable kernels on the tall 16 x 96 tile (default) and on the 64 x 16 one: the same sums in the same order
    monkeypatch.setenv('IMSEGM_SEP_WIDE_TILE', '1')
    wide = sess.lm_features(filters, D.MAX_SIGNAL_RESPONSE, mean='mean' in flags, std='std' in flags, energy='energy' in flags)
    planes_wide = [(sess.lm_battery(filters[b], D.MAX_SIGNAL_RESPONSE), sess.get_response()) for b in (1, 3, 4)]
    monkeypatch.delenv('IMSEGM_SEP_WIDE_TILE')
    assert np.array_equal(one, wide)
    for b, (norm, planes) in zip((1, 3, 4), planes_wide):              # (bar battery, Gaussian, Laplacian of a Gaussian)
        assert sess.lm_battery(filters[b], D.MAX_SIGNAL_RESPONSE) == norm and np.array_equal(sess.get_response(), planes)
    # (the two calls did take different kernels; the fixed-point statistics absorb most of the last-bit differences of the sums)
    assert sorted(set(_hip.Image2D._pack_bank(filters, True, True)['parity'].tolist())) == [-2, 0, 2]
    assert sorted(set(_hip.Image2D._pack_bank(filters, True, False)['parity'].tolist())) == [-1, 0, 1]
    sess.close()
    # ... and through the descriptor function the pipelines call
    fts, names = D.compute_texture_desc_lm_img2d_clr(img, seg, list(flags))
    assert fts.shape == one.shape and len(names) == one.shape[1]
    assert np.array_equal(fts, np.nan_to_num(one) + 0.0)


def test_images_in_flight_match_sequential():
    """worker threads (one HIP stream each) keep several images in flight: same results as one at a time"""
    from pyimsegm_amd import pipelines as pipe
    from pyimsegm_amd.utilities.synthetic import voronoi_image
    images = [voronoi_image(200, 260, seed=40 + i) for i in range(6)]
    feats = {'color': ('mean', 'std', 'energy')}
    np.random.seed(0)
    model, fts_seq = pipe.estim_model_classes_group(images, 3, feats, sp_size=15, sp_regul=0.2, nb_workers=1)
    np.random.seed(0)
    _, fts_par = pipe.estim_model_classes_group(images, 3, feats, sp_size=15, sp_regul=0.2, nb_workers=3)
    assert all(np.array_equal(a, b) for a, b in zip(fts_seq, fts_par))
    seq = [pipe.segment_color2d_slic_features_model_graphcut(im, model, feats, sp_size=15, sp_regul=0.2, gc_regul=2.)[0]
           for im in images]
    par = pipe.segment_batch_color2d_slic_features_model_graphcut(images, model, feats, sp_size=15, sp_regul=0.2,
                                                                  gc_regul=2., nb_workers=3)
    assert len(par) == len(seq)
    assert all(np.array_equal(a, b) for a, b in zip(seq, par))
    # one worker: all six images go through ONE recycled session; a second size interleaved gets its own
    one = pipe.segment_batch_color2d_slic_features_model_graphcut(images, model, feats, sp_size=15, sp_regul=0.2,
                                                                  gc_regul=2., nb_workers=1)
    assert all(np.array_equal(a, b) for a, b in zip(seq, one))
    other = voronoi_image(120, 300, seed=77)
    ref_other = pipe.segment_color2d_slic_features_model_graphcut(other, model, feats, sp_size=15, sp_regul=0.2, gc_regul=2.)[0]
    mixed = [pipe.segment_batch_color2d_slic_features_model_graphcut([im], model, feats, sp_size=15, sp_regul=0.2,
                                                                     gc_regul=2., nb_workers=1)[0]
             for im in (images[0], other, images[1], other)]
    assert np.array_equal(mixed[0], seq[0]) and np.array_equal(mixed[2], seq[1])
    assert np.array_equal(mixed[1], ref_other) and np.array_equal(mixed[3], ref_other)


@pytest.mark.parametrize('key', ['descriptors', 'graph_cuts', 'superpixels', 'classification', 'labeling', 'pipelines', 'utilities_data_io'])
def test_examples_of_the_host_modules_on_the_device(key):
    """every example of tests/doctests -- the reference's doctest vectors of the mirrored functions among them -- with the ones
    that need the GPU included"""
    import warnings
    from tests.doctests import run_examples
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        failed, attempted = run_examples(key, on_device=True)
    assert failed == 0 and attempted > 0
