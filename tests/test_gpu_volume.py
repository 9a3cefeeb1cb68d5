"""GPU parity tests of the gray-volume (supervoxel) rows of the hot path: SLIC with anisotropic
spacing, connected-component relabelling, gray statistics, 6-connected adjacency and centres, and
the whole 3D pipeline -- HIP path (through the C ABI) against the CPU oracle, bit-exact labels /
edges, floats within 1e-9, plus the reference's own doctest vectors for these functions."""
import numpy as np
import pytest

from pyimsegm_amd.utilities.synthetic import ellipsoid_volume

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def hip():
    from pyimsegm_amd import _hip
    _hip.default_context()
    return _hip


def _noisy_ellipsoid(shape, seed=0, dtype=np.float64):
    rng = np.random.default_rng(seed)
    vol = ellipsoid_volume(shape).astype(np.float64)
    vol = 0.25 + 0.5 * vol + 0.1 * rng.standard_normal(shape)
    if dtype == np.uint8:
        return np.clip(vol * 255, 0, 255).astype(np.uint8)
    if dtype == np.uint16:
        return np.clip(vol * 40000, 0, 65535).astype(np.uint16)
    return vol.astype(dtype)


def _oracle_raw(oracle, vol, n_seg, compact, space, **kw):
    """SLIC + connectivity as scikit-image 0.18 evaluates it for the dtype: float32 volumes run in float32"""
    if vol.dtype == np.float32:
        return oracle.slic_gray3d_float32(vol, n_seg, compact, sigma=1., spacing=space, **kw)
    return oracle.slic(vol, n_seg, compact, sigma=1, spacing=space, multichannel=False, **kw)


VOLUMES = [
    ('iso_f64', (24, 40, 48), np.float64, 7, 0.2, (1, 1, 1)),
    ('iso_f32', (24, 40, 48), np.float32, 7, 0.2, (1, 1, 1)),
    ('aniso_z_f32', (9, 64, 70), np.float32, 12, 0.2, (5, 1, 1)),
    ('bricks_f32', (35, 50, 200), np.float32, 8, 0.1, (2, 1, 1)),
    ('one_slice_f32', (1, 50, 60), np.float32, 8, 0.2, (1, 1, 1)),
    ('aniso_z_f64', (9, 64, 70), np.float64, 12, 0.2, (5, 1, 1)),
    ('aniso_x_u8', (50, 60, 11), np.uint8, 10, 0.3, (1, 1, 5)),
    ('ragged_u16', (7, 33, 129), np.uint16, 9, 0.25, (3, 1, 1)),
    ('one_slice', (1, 50, 60), np.float64, 8, 0.2, (1, 1, 1)),
    # supervoxels of edge 3: ~900 search windows meet one 64 x 16 cross-section, the float32 assignment walks them in several
    # batches of its 512-entry staging list; ragged in every axis
    ('many_candidates_f32', (9, 45, 150), np.float32, 3, 0.3, (1, 1, 1)),
    ('many_candidates_aniso_f32', (6, 37, 131), np.float32, 4, 0.2, (2, 1, 1)),
    ('many_candidates_u8', (9, 45, 150), np.uint8, 3, 0.3, (1, 1, 1)),
    ('many_candidates_aniso_f64', (6, 37, 131), np.float64, 4, 0.2, (2, 1, 1)),
]


@pytest.mark.parametrize('name,shape,dtype,sp,regul,space', VOLUMES, ids=[v[0] for v in VOLUMES])
def test_volume_slic_bit_exact(hip, oracle, name, shape, dtype, sp, regul, space):
    from pyimsegm_amd.superpixels import _slic3d_params
    vol = _noisy_ellipsoid(shape, dtype=dtype)
    n_seg, compact = _slic3d_params(shape, sp, regul, space)
    ref_raw = _oracle_raw(oracle, vol, n_seg, compact, space)
    sess = hip.Volume3D(*shape).upload(vol)
    sess.slic(n_seg, compact, sigma=1., spacing=space)
    raw = sess.get_labels()
    assert raw.shape == shape
    assert np.array_equal(raw, ref_raw), 'SLIC + connectivity differ in %d voxels' % np.count_nonzero(raw != ref_raw)
    k = sess.label_cc()
    ref = oracle.label_cc(ref_raw)
    got = sess.get_labels()
    assert np.array_equal(got, ref)
    assert k == ref.max() + 1
    sess.close()


def test_volume_slic_without_connectivity(hip, oracle):
    vol = _noisy_ellipsoid((12, 40, 44), seed=4)
    ref = oracle.slic(vol, 150, 3, sigma=1, spacing=(2, 1, 1), multichannel=False, enforce_connectivity=False)
    sess = hip.Volume3D(*vol.shape).upload(vol)
    sess.slic(150, 3, sigma=1., spacing=(2, 1, 1), enforce_connectivity=False)
    assert np.array_equal(sess.get_labels(), ref)
    sess.close()


@pytest.mark.parametrize('dtype', [np.float64, np.float32])
def test_volume_slic_brick_list_overflow(hip, oracle, monkeypatch, dtype):
    """bricks whose candidate list overflows scan the whole centroid table: same result"""
    vol = _noisy_ellipsoid((20, 48, 70), seed=7, dtype=dtype)
    ref = _oracle_raw(oracle, vol, 300, 2, (1, 1, 1))
    monkeypatch.setenv('IMSEGM_BRICK_CAP', '3')
    sess = hip.Volume3D(*vol.shape).upload(vol)
    sess.slic(300, 2, sigma=1., spacing=(1, 1, 1))
    assert np.array_equal(sess.get_labels(), ref)
    sess.close()


def test_float32_volume_without_connectivity_and_session_reuse(hip, oracle):
    """raw float32 k-means assignment (no connectivity pass), then a float64 volume through the same session"""
    vol = _noisy_ellipsoid((12, 40, 44), seed=4, dtype=np.float32)
    ref = _oracle_raw(oracle, vol, 150, 3, (2, 1, 1), enforce_connectivity=False)
    sess = hip.Volume3D(*vol.shape).upload(vol)
    sess.slic(150, 3, sigma=1., spacing=(2, 1, 1), enforce_connectivity=False)
    assert np.array_equal(sess.get_labels(), ref)
    vol64 = _noisy_ellipsoid((12, 40, 44), seed=5)
    sess.upload(vol64)
    sess.slic(150, 3, sigma=1., spacing=(2, 1, 1))
    assert np.array_equal(sess.get_labels(), _oracle_raw(oracle, vol64, 150, 3, (2, 1, 1)))
    sess.close()


@pytest.mark.parametrize('shape,n_seg,spacing', [((12, 40, 44), 150, (2, 1, 1)), ((9, 70, 130), 420, (1, 1, 1)), ((5, 33, 257), 90, (3, 1, 1))])
def test_float32_centroid_update_in_raster_order(hip, oracle, shape, n_seg, spacing):
    """the raster-order float32 sums of a segment by ONE LANE per supervoxel (sixty-four segments advance per wave; the form with one
    wave per segment of rounds 3 / 4 went in round 6): the k-means assignment of the oracle, i.e. of `_slic_cython[float32]`"""
    vol = _noisy_ellipsoid(shape, seed=9 + shape[0], dtype=np.float32)
    ref = _oracle_raw(oracle, vol, n_seg, 3, spacing, enforce_connectivity=False)
    sess = hip.Volume3D(*vol.shape).upload(vol)
    sess.slic(n_seg, 3, sigma=1., spacing=spacing, enforce_connectivity=False)
    assert np.array_equal(sess.get_labels(), ref)
    sess.close()


def test_segment_slic_img3d_gray_api(oracle):
    from pyimsegm_amd import superpixels as sp
    np.random.seed(0)
    img = np.random.random((100, 100, 10))           # the reference's doctest call, superpixels.py:82-86
    slic = sp.segment_slic_img3d_gray(img, 20, 0.2, (1, 1, 5))
    assert slic.shape == (100, 100, 10) and slic.dtype == np.int64
    assert np.array_equal(slic, oracle.segment_slic_img3d_gray(img, 20, 0.2, (1, 1, 5)))
    # SLIC segment 0 is background to measure.label and keeps the value 0; components count from 1
    assert slic.min() == 0 and len(np.unique(slic)) == slic.max() + 1


def test_label_cc_background_and_diagonals(hip, oracle):
    rng = np.random.default_rng(2)
    lab = rng.integers(0, 3, (9, 17, 70))
    sess = hip.Volume3D(*lab.shape).set_labels(lab)
    k = sess.label_cc()
    ref = oracle.label_cc(lab)
    assert np.array_equal(sess.get_labels(), ref)
    assert k == ref.max() + 1
    assert np.all((ref == 0) == (lab == 0))
    sess.close()


@pytest.mark.parametrize('full', [False, True])
def test_label_cc_by_runs_and_by_every_neighbour(hip, oracle, monkeypatch, full):
    """round 5: measure.label's merge pass ties runs (a few unions where a run starts or the row above changes); the pass with
    thirteen unions per voxel stays behind IMSEGM_CC_MERGE_FULL -- both against the oracle: random labels (contacts through edges
    and corners only), blocky labels (long runs), thin volumes, a single row"""
    if full:
        monkeypatch.setenv('IMSEGM_CC_MERGE_FULL', '1')
    rng = np.random.default_rng(12)
    blocks = np.kron(rng.integers(0, 4, (3, 5, 9)), np.ones((4, 6, 13), dtype=np.int64))
    blocks[rng.random(blocks.shape) < 0.08] = 0
    cases = [rng.integers(0, 3, (9, 17, 70)), rng.integers(0, 2, (1, 40, 131)), rng.integers(0, 4, (6, 1, 300)), blocks,
             rng.integers(1, 3, (3, 3, 257)), (rng.random((7, 33, 65)) < 0.3).astype(np.int64) * 5]
    for lab in cases:
        sess = hip.Volume3D(*lab.shape).set_labels(lab)
        k = sess.label_cc()
        ref = oracle.label_cc(lab)
        assert np.array_equal(sess.get_labels(), ref) and k == ref.max() + 1
        sess.close()


@pytest.mark.parametrize('general', [False, True])
def test_label_cc_of_the_connectivity_pass_by_first_voxels(hip, oracle, monkeypatch, general):
    """round 6: every label > 0 the connectivity pass writes is one 6-connected set, so measure.label of ITS map only renumbers the
    labels by their first voxels (connectivity.hip launch_label_connected); the union-find of every other map stays behind
    IMSEGM_LABEL_GENERAL -- both against the oracle: tiny supervoxels in noise (most components small and merged), truncated
    components (max_size_factor 1.1), start_label 1, one slice; and a map set by the caller afterwards (a label in two pieces) must
    take the union-find again"""
    if general:
        monkeypatch.setenv('IMSEGM_LABEL_GENERAL', '1')
    rng = np.random.default_rng(31)
    cases = [((8, 40, 130), np.float32, 900, 0.05, (1, 1, 1), {}),
             ((9, 45, 67), np.float64, 500, 0.02, (2, 1, 1), {}),
             ((6, 50, 70), np.float64, 40, 5.0, (3, 1, 1), {'max_size_factor': 1.1, 'min_size_factor': 0.2}),
             ((5, 33, 129), np.uint8, 120, 0.5, (1, 1, 1), {'start_label': 1}),
             ((1, 64, 90), np.float32, 60, 1.0, (1, 1, 1), {})]
    for shape, dtype, n_seg, compact, spacing, kw in cases:
        vol = rng.random(shape)
        vol = (vol * 255).astype(np.uint8) if dtype == np.uint8 else vol.astype(dtype)
        ref_raw = _oracle_raw(oracle, vol, n_seg, compact, spacing, **kw)
        sess = hip.Volume3D(*shape).upload(vol)
        sess.slic(n_seg, compact, sigma=1., spacing=spacing, **kw)
        assert np.array_equal(sess.get_labels(), ref_raw)
        k = sess.label_cc()
        ref = oracle.label_cc(ref_raw)
        assert np.array_equal(sess.get_labels(), ref) and k == ref.max() + 1, (shape, kw)
        k2 = sess.label_cc()                          # (the relabelled map is of the same kind: numbering it again changes nothing)
        assert np.array_equal(sess.get_labels(), ref) and k2 == k
        # the caller's own map through the same session: label 2 in two pieces
        lab = np.ones(shape, dtype=np.int64)
        lab[..., :3] = 2
        lab[..., -3:] = 2
        sess.set_labels(lab)
        sess.label_cc()
        assert np.array_equal(sess.get_labels(), oracle.label_cc(lab))
        sess.close()


def test_gray_statistics(hip, oracle):
    from pyimsegm_amd import descriptors as d
    for dtype in (np.float64, np.uint8):
        vol = _noisy_ellipsoid((10, 37, 66), seed=1, dtype=dtype)
        seg = oracle.segment_slic_img3d_gray(vol, 9, 0.2, (2, 1, 1))
        mean = d.cython_img3d_gray_mean(vol, seg)
        energy = d.cython_img3d_gray_energy(vol, seg)
        std = d.cython_img3d_gray_std(vol, seg)
        v32 = np.array(vol, dtype=np.float32)
        s32 = np.array(seg, dtype=np.int32)
        ref_mean = oracle.gray3d_stat(v32, s32, 'mean')
        ref_energy = oracle.gray3d_stat(v32, s32, 'energy')
        ref_var = oracle.gray3d_stat(v32, s32, 'var', np.array(ref_mean, dtype=np.float32))
        np.testing.assert_allclose(mean, ref_mean, rtol=1e-12, atol=1e-12)
        np.testing.assert_allclose(energy, ref_energy, rtol=1e-12, atol=1e-12)
        np.testing.assert_allclose(std, np.sqrt(ref_var), rtol=1e-9, atol=1e-9)
        ref = oracle.ref_features_cython()
        if ref is not None:                            # the reference's own compiled Cython, when built
            np.testing.assert_allclose(mean, np.asarray(ref.computeGrayImage3dMean(v32, s32)), rtol=1e-5, atol=1e-5)
            np.testing.assert_allclose(energy, np.asarray(ref.computeGrayImage3dEnergy(v32, s32)), rtol=1e-5,
                                       atol=1e-5)


def test_reference_doctest_vectors_gray3d():
    """descriptors.py:471-478, 502-508, 532-538, 715-740 of the reference"""
    from pyimsegm_amd import descriptors as d
    image = np.zeros((2, 3, 8))
    image[0, :, 2:6] = 1
    image[1, :, 3:7] = 3
    segm = np.array([[[0, 0, 0, 0, 1, 1, 1, 1]] * 3, [[2, 2, 2, 2, 3, 3, 3, 3]] * 3])
    assert d.cython_img3d_gray_mean(image, segm).tolist() == [0.5, 0.5, 0.75, 2.25]
    assert d.cython_img3d_gray_energy(image, segm).tolist() == [0.5, 0.5, 2.25, 6.75]
    np.testing.assert_allclose(d.cython_img3d_gray_std(image, segm), [0.5, 0.5, 1.29903811, 1.29903811], atol=1e-8)
    segm5 = np.array([[[0, 0, 0, 0, 1, 1, 1, 1]] * 3, [[2, 2, 2, 2, 5, 5, 5, 5]] * 3])
    features, names = d.compute_image3d_gray_statistic(image, segm5)
    assert names == ['gray_mean', 'gray_std', 'gray_energy', 'gray_median', 'gray_meanGrad']
    expect = [[0.5, 0.5, 0.5, 0.5, 0.25], [0.5, 0.5, 0.5, 0.5, -0.25], [0.75, 1.299, 2.25, 0., 0.75],
              [0., 0., 0., 0., 0.], [0., 0., 0., 0., 0.], [2.25, 1.299, 6.75, 3., -1.125]]
    np.testing.assert_allclose(np.round(features, 3), expect, atol=1e-12)
    np.random.seed(0)
    img = np.random.random((2, 10, 15))
    slic = np.zeros((2, 10, 15), dtype=int)
    slic[:, :, :7] += 1
    slic[1, :, :] += 2
    fts, names = d.compute_selected_features_gray3d(img, slic, {'color': ('mean', 'std', 'median')})
    assert fts.shape == (4, 3) and names == ['gray_mean', 'gray_std', 'gray_median']
    np.testing.assert_allclose(fts[:, 0], [img[slic == i].astype(np.float32).mean(dtype=np.float64) for i in range(4)],
                               atol=1e-7)


@pytest.mark.parametrize('shape,space', [((24, 40, 48), (1, 1, 1)), ((9, 64, 70), (5, 1, 1)), ((35, 50, 200), (2, 1, 1)),
                                         ((1, 50, 60), (1, 1, 1)), ((3, 5, 7), (1, 1, 1)), ((20, 33, 130), (1, 2, 3))])
def test_float32_blur_by_columns_and_tiles_equals_the_three_axis_passes(hip, monkeypatch, shape, space):
    """round 6: the Gaussian of a float32 volume as a z pass over register windows + a fused y / x pass through LDS
    (volume.hip k_vol_blur_z32 / k_vol_blur_yx32) against the three axis passes of rounds 2 - 5 (IMSEGM_PRE_3PASS): the k-means
    assignment and the supervoxel map after it do not move by a voxel -- ragged sizes, a volume smaller than the filter, one
    slice, a different radius per axis (sigma / spacing), W not a multiple of four (scalar loads)"""
    from pyimsegm_amd.superpixels import _slic3d_params
    vol = _noisy_ellipsoid(shape, seed=9, dtype=np.float32)
    n_seg, compact = _slic3d_params(vol.shape, 6, 0.2, space)
    n_seg, compact, spacing = max(n_seg, 2), max(compact, 1), space
    maps = []
    for three_pass in (False, True):
        if three_pass:
            monkeypatch.setenv('IMSEGM_PRE_3PASS', '1')
        sess = hip.Volume3D(*vol.shape).upload(vol)
        sess.slic(n_seg, compact, sigma=1., spacing=spacing, enforce_connectivity=False)
        raw = sess.get_labels()
        sess.slic(n_seg, compact, sigma=1., spacing=spacing)
        maps.append((raw, sess.get_labels()))
        sess.close()
    assert np.array_equal(maps[0][0], maps[1][0]) and np.array_equal(maps[0][1], maps[1][1])


def test_reference_doctest_vectors_graph3d():
    """superpixels.py:185-192 and :214-215 of the reference"""
    from pyimsegm_amd import superpixels as sp
    grid_2d = np.array([[0] * 5 + [1] * 5, [2] * 5 + [3] * 5])
    grid = np.array([grid_2d, grid_2d + 4])
    v, edges = sp.make_graph_segm_connect_grid3d_conn6(grid)
    assert v.tolist() == [0, 1, 2, 3, 4, 5, 6, 7]
    assert edges == [[0, 1], [0, 2], [1, 3], [2, 3], [0, 4], [1, 5], [4, 5], [2, 6], [4, 6], [3, 7], [5, 7], [6, 7]]
    segm = np.array([[0] * 6 + [1] * 5, [0] * 6 + [2] * 5])
    assert sp.superpixel_centers(np.array([segm, segm, segm])) == [[1.0, 0.5, 2.5], [1.0, 0.0, 8.0], [1.0, 1.0, 8.0]]


def test_volume_graph_vs_oracle(hip, oracle):
    vol = _noisy_ellipsoid((14, 45, 52), seed=5)
    seg = oracle.segment_slic_img3d_gray(vol, 8, 0.2, (3, 1, 1))
    sess = hip.Volume3D(*seg.shape).set_labels(seg)
    edges, centres, present = sess.graph()
    ref_v, ref_e = oracle.adjacency(seg)
    assert np.flatnonzero(present).tolist() == ref_v.tolist()
    assert edges.tolist() == ref_e
    ref_c = oracle.centers(seg)
    assert np.array_equal(centres[present], ref_c[present])       # exact integer sums / counts on both sides
    assert np.all(centres[~present] == -1)
    sess.close()


def test_volume_graph_from_the_neighbour_table(hip, oracle, monkeypatch):
    """the graph of a label volume with the neighbours of a label kept in a table of slots per label instead of a K x K bitmap
    (what volumes of more than 46 000 labels take: the 298 116 supervoxels of BASELINE configs[4]; here forced): the edges of the
    oracle in the oracle's order -- also with one label that touches 150 labels of smaller number (rows widened 32 -> 256)"""
    monkeypatch.setenv('IMSEGM_ADJACENCY_TABLE', '1')
    vol = _noisy_ellipsoid((14, 45, 52), seed=5)
    seg = oracle.segment_slic_img3d_gray(vol, 8, 0.2, (3, 1, 1))
    wide = (np.arange(6 * 30 * 40).reshape(6, 30, 40) // 8) % 150           # 150 labels in runs of 8 voxels ...
    wide[3] = 150                                                           # ... and a slab between them that touches them all
    for labels in (seg, wide.astype(np.int64)):
        sess = hip.Volume3D(*labels.shape).set_labels(labels)
        edges, centres, present = sess.graph()
        ref_v, ref_e = oracle.adjacency(labels)
        assert np.flatnonzero(present).tolist() == ref_v.tolist()
        assert edges.tolist() == ref_e
        assert np.array_equal(centres[present], oracle.centers(labels)[present])
        sess.close()
    assert sum(1 for a, b in ref_e if b == 150) == 150


def test_pipe_gray3d(oracle):
    """reference doctest pipelines.py:402-407 + stage-by-stage equality with the oracle"""
    from pyimsegm_amd import pipelines, descriptors as d, graph_cuts as gc
    np.random.seed(0)
    image = np.random.random((5, 125, 150)) / 2.
    image[:, :, :75] += 0.5
    np.random.seed(0)
    segm = pipelines.pipe_gray3d_slic_features_model_graphcut(image, 2, {'color': ['mean']})
    assert segm.shape == (5, 125, 150)
    # the same pipeline with every native stage taken from the oracle and the identical host glue
    slic = oracle.segment_slic_img3d_gray(image, 15, 0.2, (12, 1, 1))
    mean = oracle.gray3d_stat(np.array(image, dtype=np.float32), slic.astype(np.int32), 'mean')
    features = np.nan_to_num(mean[:, np.newaxis])
    features, _ = d.norm_features(features)
    np.random.seed(0)
    model = gc.estim_class_model(features, 2)
    proba = model.predict_proba(features)
    _, edges = oracle.adjacency(slic)
    edges = np.array(edges, dtype=np.int32)
    weights = gc.compute_edge_model(edges, proba, 'lT')
    weights = weights / gc.compute_spatial_dist(oracle.centers(slic), edges, relative=True)
    weights = np.clip(weights, 1e-3, 1e3)
    labels = oracle.cut_general_graph(edges, weights, gc.compute_unary_cost(proba),
                                      gc.compute_pairwise_cost(0.1, proba.shape), n_iter=-1)
    assert np.array_equal(segm, np.asarray(labels)[slic])
    # the two halves of the volume end up in different classes
    left, right = segm[:, :, :60], segm[:, :, 90:]
    assert np.mean(left == np.bincount(left.ravel()).argmax()) > 0.95
    assert np.bincount(left.ravel()).argmax() != np.bincount(right.ravel()).argmax()


def test_pipe_gray3d_when_the_device_has_no_room_for_the_bit_arrays(hip, monkeypatch):
    """the fused graph / terms / cut call needs two K x K bit arrays; the library asks the device what is free and
    refuses when they do not fit (here forced by IMSEGM_FUSED_BITMAP_MB = 1 MB against 2 x 3.4 MB at 5 270 supervoxels), and the
    pipeline falls back to the neighbour-table graph + imsegm_cut_general_graph -- same segmentation"""
    from pyimsegm_amd import pipelines
    rng = np.random.default_rng(3)
    image = rng.random((12, 200, 220)) / 2.
    image[:, :, :110] += 0.5
    np.random.seed(0)
    fused = pipelines.pipe_gray3d_slic_features_model_graphcut(image, 2, {'color': ['mean', 'std']}, sp_size=10, spacing=(1, 1, 1))
    monkeypatch.setenv('IMSEGM_FUSED_BITMAP_MB', '1')
    sess = hip.Volume3D(*image.shape).set_labels((np.arange(image.size).reshape(image.shape) // 7 % 6000).astype(np.int64))
    with pytest.raises(hip.HipFusedPathError, match='fused path'):          # (status IMSEGM_E_FUSED_PATH, not a message match)
        sess.segment(np.zeros((2, 2)), 'model', proba=np.full((6000, 2), 0.5), pinned=False)
    sess.close()
    np.random.seed(0)
    by_tables = pipelines.pipe_gray3d_slic_features_model_graphcut(image, 2, {'color': ['mean', 'std']}, sp_size=10, spacing=(1, 1, 1))
    assert np.array_equal(fused, by_tables) and len(np.unique(fused)) == 2


def test_all_finite_and_session_reuse(hip):
    """imsegm_image2d_all_finite on the uploaded pixels (float32 / float64 volumes, a float64 colour image, uint8), and the
    pipeline on a recycled volume session: the second volume of a shape gives what a fresh session gives"""
    from pyimsegm_amd import pipelines
    rng = np.random.default_rng(5)
    for dtype in (np.float32, np.float64):
        vol = rng.random((3, 40, 50)).astype(dtype)
        sess = hip.Volume3D(*vol.shape).upload(vol)
        assert sess.all_finite() is True
        for bad in (np.nan, np.inf, -np.inf):
            v2 = vol.copy()
            v2[2, 39, 49] = bad
            assert sess.upload(v2).all_finite() is False
        assert sess.upload(vol).all_finite() is True
        sess.close()
    img = rng.random((30, 41, 3))
    sess = hip.Image2D(30, 41)
    sess.upload(img)
    assert sess.all_finite() is True
    img[29, 40, 2] = np.nan
    sess.upload(img)
    assert sess.all_finite() is False
    sess.upload(np.zeros((30, 41, 3), dtype=np.uint8))
    assert sess.all_finite() is True
    sess.close()
    vols = []
    for seed in (1, 2):
        v = np.random.default_rng(seed).random((4, 60, 70)) / 2.
        if seed == 1:
            v[:, :, :35] += 0.5
        else:
            v[:, :25, :] += 0.5
        vols.append(v)
    out = []
    for v in (vols[0], vols[1], vols[0]):                 # the third call runs on the session the second one left
        np.random.seed(0)
        out.append(pipelines.pipe_gray3d_slic_features_model_graphcut(v, 2, {'color': ['mean', 'std']}, spacing=(2, 1, 1), sp_size=10))
    assert (4, 60, 70) in hip.default_context().idle_sessions
    assert np.array_equal(out[0], out[2]) and not np.array_equal(out[0], out[1])
    # volumes of another shape back to back: at most ONE idle volume session per context (a volume session owns
    # ~70 bytes of device memory per voxel; one idle session per distinct shape ended in hipMalloc failures)
    for shape in ((3, 50, 64), (5, 40, 48)):
        np.random.seed(0)
        w = np.random.default_rng(9).random(shape) / 2.
        w[:, :, :shape[2] // 2] += 0.5
        assert pipelines.pipe_gray3d_slic_features_model_graphcut(w, 2, {'color': ['mean']}, spacing=(2, 1, 1), sp_size=10).shape == shape
        idle3 = [k for k in hip.default_context().idle_sessions if len(k) == 3]
        assert idle3 == [shape], idle3
    v = vols[0].copy()
    v[0, 0, 0] = np.nan                                    # non-finite voxels: the general path (descriptors on the host)
    np.random.seed(0)
    assert pipelines.pipe_gray3d_slic_features_model_graphcut(v, 2, {'color': ['mean']}, spacing=(2, 1, 1), sp_size=10).shape == v.shape


def test_volume_slic_randomised_sweep(hip, oracle):
    """seeded sweep over volume shapes, dtypes, spacings, supervoxel sizes and compactness (colour- to
    space-dominated): raw SLIC + connectivity and the measure.label relabelling, bit for bit"""
    from pyimsegm_amd.superpixels import _slic3d_params
    rng = np.random.default_rng(77)
    for case in range(12):
        shape = (int(rng.integers(1, 20)), int(rng.integers(20, 70)), int(rng.integers(20, 90)))
        dtype = [np.float64, np.uint8, np.float32, np.uint16][case % 4]
        vol = _noisy_ellipsoid(shape, seed=int(rng.integers(1000)), dtype=np.float64 if dtype == np.float32 else dtype)
        if dtype == np.float32:
            vol = vol.astype(np.float32)
        space = [(1, 1, 1), (3, 1, 1), (1, 1, 2), (6, 1, 1)][int(rng.integers(4))]
        sp = int(rng.integers(5, 14))
        regul = float(rng.choice([0.05, 0.2, 0.6]))
        n_seg, compact = _slic3d_params(shape, sp, regul, space)
        if n_seg < 1 or compact < 1:
            continue
        ref_raw = _oracle_raw(oracle, vol, n_seg, compact, space)
        sess = hip.Volume3D(*shape).upload(vol)
        sess.slic(n_seg, compact, sigma=1., spacing=space)
        raw = sess.get_labels()
        assert np.array_equal(raw, ref_raw), 'case %d %r %s spacing %r sp %d regul %g: %d voxels differ' % (
            case, shape, np.dtype(dtype).name, space, sp, regul, np.count_nonzero(raw != ref_raw))
        sess.label_cc()
        assert np.array_equal(sess.get_labels(), oracle.label_cc(ref_raw))
        sess.close()
