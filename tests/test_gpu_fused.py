"""The fused back half of the pipeline (imsegm_image2d_features_color + imsegm_image2d_segment: feature table, class
model, unary / edge terms, pyGCO integers, CSR, alpha-expansion, gathers in one enqueue) against
  * scikit-learn's own predict_proba (1e-9) and the host mirror functions of graph_cuts (1e-12),
  * the integer energies pygco would form from those terms (identical),
  * the stage-by-stage path of round 1 (separate C calls with host numpy in between): same labels, same segmentation,
  * the vectors of the reference's own run (tests/golden/reference.npz): the integers it handed to gco, its labels."""
import numpy as np
import pytest

from pyimsegm_amd.utilities.synthetic import ellipsoid_volume, voronoi_image
from test_golden_reference import GEN, NAMES, VEC, make_input, rebuild_model

pytestmark = pytest.mark.gpu


def pygco_integers(unary, weights, pairwise):
    """float -> int conversion of gco-wrapper's pygco.cut_general_graph (restated in api.hip for the stand-alone call)"""
    mu, mw, mp = np.abs(unary).max(), np.abs(weights).max() if len(weights) else 0., pairwise.max()
    dwf = (mw * mp if (len(weights) and mw * mp > mu) else mu) + 1e-10
    return ((unary / dwf) * 100000).astype(np.int32), ((weights / dwf) * 1000).astype(np.int32)


@pytest.fixture(scope='module')
def fitted():
    from pyimsegm_amd import _hip
    from pyimsegm_amd import graph_cuts as G
    from pyimsegm_amd import superpixels as S
    image = voronoi_image(300, 400, seed=3)
    sess, mode = S._open_session(image)
    S._run_slic(sess, mode, 20, 0.2)
    features = sess.features_color(True, True, True, to_host=True)
    np.random.seed(0)
    model = G.estim_class_model(features, 3, 'GMM', None, True)
    yield image, sess, features, model, _hip.DeviceGmm(model)
    sess.close()


def test_feature_table_equals_the_separate_statistics(fitted):
    image, sess, features, _, _ = fitted
    mean, energy, var = sess.color_stats()
    assert np.array_equal(features, np.hstack([mean, np.sqrt(var), energy]))
    assert np.array_equal(sess.features_color(True, False, True), np.hstack([mean, energy]))
    assert np.array_equal(sess.features_color(False, True, False), np.sqrt(var))
    sess.features_color(True, True, True, to_host=False)


@pytest.mark.parametrize('edge_type', ['model', 'model_l1', 'model_l2', 'model_lT', 'spatial', 'features', '', 'const'])
def test_fused_terms_equal_sklearn_and_the_host_mirror(fitted, edge_type):
    from pyimsegm_amd import _hip
    from pyimsegm_amd import graph_cuts as G
    image, sess, features, model, gmm = fitted
    pairwise = G.compute_pairwise_cost(1.5, (len(features), 3))
    out = sess.segment(pairwise, edge_type, edge_cost=1.0, gmm=gmm, debug=True, want_soft=True)
    proba = out['proba']
    np.testing.assert_allclose(proba, model.predict_proba(features), rtol=0, atol=1e-9)
    np.testing.assert_allclose(proba, G.predict_proba(model, features), rtol=0, atol=1e-9)
    edges, centres, _ = sess.graph()
    assert np.array_equal(out['edges'], edges) and np.array_equal(out['centres'], centres)
    # terms from the DEVICE probabilities through the host functions of the reference mirror
    np.testing.assert_allclose(out['unary'], G.compute_unary_cost(proba), rtol=1e-13, atol=1e-13)
    ref_w = G.edge_weights_from_graph(edges, centres, features, proba, edge_type)
    np.testing.assert_allclose(out['edge_weights'], ref_w, rtol=1e-10, atol=1e-13)
    ui, wi = pygco_integers(out['unary'], out['edge_weights'], pairwise)
    assert np.array_equal(out['unary_int'], ui) and np.array_equal(out['edge_weights_int'], wi)
    # the graph cut on those very terms through the stand-alone entry point (host-built CSR) gives the same labels
    labels, energy = _hip.cut_general_graph(edges, out['edge_weights'], out['unary'], pairwise, return_energy=True)
    assert np.array_equal(out['graph_labels'], labels) and out['energy'] == energy
    slic = sess.get_labels()
    assert out['segm'].dtype == np.int32 and np.array_equal(out['segm'], labels[slic])
    assert np.array_equal(out['soft'], proba[slic])


def test_fused_variants(fitted):
    """probabilities from the host, classes_ LUT, gc_regul <= 0, edge_cost, pageable outputs, soft kept on the device"""
    from pyimsegm_amd import graph_cuts as G
    image, sess, features, model, gmm = fitted
    slic = sess.get_labels()
    proba = model.predict_proba(features)
    pairwise = G.compute_pairwise_cost(2.0, proba.shape)
    a = sess.segment(pairwise, 'model', gmm=gmm, want_graph_labels=True)
    b = sess.segment(pairwise, 'model', proba=proba, want_graph_labels=True, pinned=False)
    assert np.array_equal(a['graph_labels'], b['graph_labels']) and np.array_equal(a['segm'], b['segm'])
    classes = np.array([7, 3, 11])
    c = sess.segment(pairwise, 'model', gmm=gmm, classes=classes)
    assert np.array_equal(c['segm'], classes[a['graph_labels']][slic])
    d = sess.segment(G.compute_pairwise_cost(0., proba.shape), 'model', gmm=gmm, use_graphcut=False, want_graph_labels=True, debug=True)
    assert np.array_equal(d['graph_labels'], np.argmin(d['unary'], axis=-1))
    e = sess.segment(pairwise, 'model', edge_cost=3.0, gmm=gmm, debug=True)
    f = sess.segment(pairwise, 'model', edge_cost=1.0, gmm=gmm, debug=True)
    np.testing.assert_allclose(e['edge_weights'], 3.0 * f['edge_weights'], rtol=1e-15)
    g = sess.segment(pairwise, 'model', gmm=gmm, want_segm=False, keep_soft_on_device=True)
    assert g == {}
    with pytest.raises(ValueError):
        sess.segment(pairwise, 'color', gmm=gmm)


@pytest.mark.parametrize('name', NAMES)
def test_fused_path_hands_gco_the_integers_of_the_reference_run(name):
    """label map, probabilities and model of the reference's own run: the device forms the same graph, the same integer
    energies as pygco would from the reference's float terms, and the same labels"""
    from pyimsegm_amd import _hip
    from pyimsegm_amd import graph_cuts as G
    _, sp, rc, feats, nb_classes, gc_regul, edge_type = GEN.CASES[name]
    image = make_input(name)
    slic = VEC[name + '_slic']
    sess = _hip.Image2D(*slic.shape).upload(image).set_labels(slic)
    flags = feats['color']
    sess.features_color('mean' in flags, 'std' in flags, 'energy' in flags, to_host=False)
    pairwise = G.compute_pairwise_cost(gc_regul, VEC[name + '_proba'].shape)
    assert np.array_equal(pairwise, VEC[name + '_gc_pairwise'])
    # (a) the reference's probabilities handed in
    out = sess.segment(pairwise, edge_type, proba=VEC[name + '_proba'], debug=True)
    assert np.array_equal(out['edges'], VEC[name + '_gc_edges'])
    np.testing.assert_allclose(out['edge_weights'], VEC[name + '_gc_edge_weights'], rtol=1e-12, atol=1e-15)
    np.testing.assert_allclose(out['unary'], VEC[name + '_gc_unary'], rtol=1e-13, atol=1e-13)
    ui, wi = pygco_integers(VEC[name + '_gc_unary'], VEC[name + '_gc_edge_weights'], pairwise)
    assert np.array_equal(out['unary_int'], ui) and np.array_equal(out['edge_weights_int'], wi)
    assert np.array_equal(out['graph_labels'], VEC[name + '_graph_labels'])
    assert np.array_equal(out['segm'], VEC[name + '_segm'])
    # (b) the reference's fitted model evaluated on the device (descriptors of the HIP path: last-bit differences from
    # the reference's -ffast-math sums are allowed to move an integer energy by one unit)
    gmm = _hip.DeviceGmm(rebuild_model(name))
    out = sess.segment(pairwise, edge_type, gmm=gmm, debug=True)
    np.testing.assert_allclose(out['proba'], VEC[name + '_proba'], rtol=0, atol=1e-9)
    assert np.abs(out['unary_int'].astype(np.int64) - ui).max() <= 1
    assert np.mean(out['segm'] != VEC[name + '_segm']) < 1e-3
    sess.close()


def test_fused_volume_pipeline_equals_the_staged_path():
    """3-D session: 6-connected graph, centres with three coordinates, probabilities from the host"""
    from pyimsegm_amd import _hip
    from pyimsegm_amd import graph_cuts as G
    from pyimsegm_amd import superpixels as S
    rng = np.random.default_rng(3)
    vol = ellipsoid_volume((12, 40, 48)).astype(np.float64) + 0.1 * rng.standard_normal((12, 40, 48))
    sess = S._open_volume(vol)
    S._run_slic3d(sess, 8, 0.2, (2, 1, 1))
    k = sess.n_labels
    proba = rng.dirichlet(np.ones(3), size=k)
    pairwise = G.compute_pairwise_cost(0.3, proba.shape)
    out = sess.segment(pairwise, 'model', proba=proba, debug=True, pinned=False)
    edges, centres, _ = sess.graph()
    assert np.array_equal(out['edges'], edges) and np.array_equal(out['centres'], centres)
    ref_w = G.edge_weights_from_graph(edges, centres, None, proba, 'model')
    np.testing.assert_allclose(out['edge_weights'], ref_w, rtol=1e-10, atol=1e-13)
    labels = _hip.cut_general_graph(edges, out['edge_weights'], out['unary'], pairwise)
    assert np.array_equal(out['graph_labels'], labels)
    assert np.array_equal(out['segm'], labels[sess.get_labels()])
    feats = sess.features_color(True, True, True)
    m, e, v = sess.gray_stats()
    assert np.array_equal(feats[:, 0], m) and np.array_equal(feats[:, 3], np.sqrt(v)) and np.array_equal(feats[:, 6], e)
    sess.close()


def test_fused_call_on_131_072_supervoxels():
    """a label volume of 32 x 64 x 64 blocks (K^2 / 8 = 2.1 GB per bit array: beyond the 2 GB the fused call stopped at until round 4;
    BASELINE configs[4] has 298 116): graph, terms and the cut by the whole device in ONE call equal the graph call + the
    terms of the host mirror + imsegm_cut_general_graph"""
    from pyimsegm_amd import _hip
    from pyimsegm_amd import graph_cuts as G
    rng = np.random.default_rng(17)
    blocks = np.arange(32 * 64 * 64, dtype=np.int64).reshape(32, 64, 64)
    labels = np.repeat(np.repeat(np.repeat(blocks, 2, axis=0), 3, axis=1), 3, axis=2)        # 64 x 192 x 192 voxels
    k = int(labels.max()) + 1
    assert float(k)**2 / 8 > 2e9
    region = (np.arange(k) // (64 * 64 * 8)) % 3                                              # slabs of classes along z
    proba = np.full((k, 3), 0.2)
    proba[np.arange(k), region] = 0.6
    proba *= rng.uniform(0.5, 1.5, proba.shape)
    proba /= proba.sum(axis=1, keepdims=True)
    pairwise = G.compute_pairwise_cost(0.5, proba.shape)
    sess = _hip.Volume3D(*labels.shape).set_labels(labels)
    try:
        out = sess.segment(pairwise, 'model', proba=proba, debug=True, pinned=False)
        edges, centres, _ = sess.graph()
        assert np.array_equal(out['edges'], edges) and np.array_equal(out['centres'], centres)
        assert len(edges) == 31 * 64 * 64 + 32 * 63 * 64 * 2
        ref_w = G.edge_weights_from_graph(edges, centres, None, proba, 'model')
        np.testing.assert_allclose(out['edge_weights'], ref_w, rtol=1e-10, atol=1e-13)
        cut = _hip.cut_general_graph(edges, out['edge_weights'], out['unary'], pairwise)
        assert np.array_equal(out['graph_labels'], cut) and len(np.unique(cut)) == 3
        assert np.array_equal(out['segm'][::2, ::3, ::3], cut[blocks])
    finally:
        sess.close()


@pytest.mark.parametrize('edge_type', ['model', 'model_l1', 'model_l2', 'spatial', 'const'])
def test_terms_of_a_volume_by_the_whole_device_keep_the_bits_of_the_one_workgroup(monkeypatch, edge_type):
    """round 6: from 16 384 supervoxels on the graph-cut terms are computed by the whole device (terms.hip k_terms_elem ...); the sums
    keep the order of the one workgroup that serves images (k_gc_terms, behind IMSEGM_TERMS_ONE_WORKGROUP for volumes): weights,
    unary costs and their integers bit for bit, and the same labelling"""
    from pyimsegm_amd import _hip
    from pyimsegm_amd import graph_cuts as G
    rng = np.random.default_rng(23)
    blocks = np.arange(20 * 32 * 32, dtype=np.int64).reshape(20, 32, 32)
    labels = np.repeat(np.repeat(np.repeat(blocks, 2, axis=0), 3, axis=1), 2, axis=2)
    k = int(labels.max()) + 1
    assert k >= 16384
    proba = rng.dirichlet(np.ones(3) * 0.7, size=k)
    pairwise = G.compute_pairwise_cost(0.4, proba.shape)
    outs = []
    for one in (False, True):
        if one:
            monkeypatch.setenv('IMSEGM_TERMS_ONE_WORKGROUP', '1')
        sess = _hip.Volume3D(*labels.shape).set_labels(labels)
        try:
            outs.append(sess.segment(pairwise, edge_type, proba=proba, debug=True, pinned=False))
        finally:
            sess.close()
    wide, one = outs
    for key in ('edge_weights', 'edge_weights_int', 'unary', 'unary_int', 'graph_labels', 'segm'):
        assert np.array_equal(wide[key], one[key]), key
    ref_w = G.edge_weights_from_graph(wide['edges'], wide['centres'], None, proba, edge_type)
    np.testing.assert_allclose(wide['edge_weights'], ref_w, rtol=1e-10, atol=1e-13)


def test_pinned_arrays_are_recycled():
    from pyimsegm_amd import _hip
    a = _hip.pinned_empty((300, 400), np.int32)
    a[:] = 7
    addr = a.ctypes.data
    del a
    b = _hip.pinned_empty((300, 400), np.int32)
    assert b.ctypes.data == addr and b.shape == (300, 400)
    image = _hip.pinned_empty((64, 96, 3), np.uint8)
    image[...] = voronoi_image(64, 96, seed=2)
    from pyimsegm_amd import superpixels as S
    assert np.array_equal(S.segment_slic_img2d(image, 12, 0.2), S.segment_slic_img2d(np.array(image), 12, 0.2))


def test_rccl_single_rank_round_trip(monkeypatch):
    """the ctypes binding of RCCL with one rank: unique id, communicator, grouped send / recv to itself on the context's
    stream through DeviceGather (all the N > 1 gather does, minus the peers)"""
    from pyimsegm_amd import _hip
    from pyimsegm_amd.distributed import DeviceGather, Group
    monkeypatch.setenv('RANK', '0')
    monkeypatch.setenv('WORLD_SIZE', '1')
    monkeypatch.setenv('LOCAL_RANK', '0')
    group = Group()
    try:
        assert group.backend == 'rccl', group.rccl_error
        ctx = _hip.default_context()
        data = _hip.pinned_empty((2, 5, 1000), np.int32)
        data[...] = np.arange(10000).reshape(2, 5, 1000)
        back = _hip.pinned_empty((5, 1000), np.int32)
        gather = DeviceGather(group, 4000, 5, ctx)
        import ctypes as C
        dev = C.c_void_p()
        _hip._check(_hip.load_library().imsegm_device_alloc(0, data.nbytes, C.byref(dev)))
        ctx.copy(dev.value, data.ctypes.data, data.nbytes)
        for rnd in range(2):
            for item in range(5):
                gather.stage(rnd, item, dev.value + (rnd * 5 + item) * 4000, ctx)
            gather.flush(rnd)
            ctx.copy(back.ctypes.data, gather.recv, back.nbytes)
            assert np.array_equal(back, data[rnd])
        gather.close()
        _hip.load_library().imsegm_device_free(dev)
    finally:
        group.close()


def test_fused_call_takes_label_maps_that_are_no_planar_partition():
    """a user segmentation whose regions are scattered pixels (installed with set_labels) has far more neighbour pairs than the
    3 K of a planar graph: the fused call sizes its edge table again from what the device reports, and equals the staged path
    (reference: imsegm/graph_cuts.py:660-747 accepts any segmentation)"""
    from pyimsegm_amd import _hip
    from pyimsegm_amd.graph_cuts import compute_pairwise_cost, segment_graph_cut_general
    rng = np.random.default_rng(23)
    K, C = 40, 3
    labels = rng.integers(0, K, (60, 80)).astype(np.int32)           # salt: nearly every pair of labels touches
    proba = rng.dirichlet(np.ones(C), K)
    sess = _hip.Image2D(60, 80).set_labels(labels, K)
    try:
        res = sess.segment(compute_pairwise_cost(1., proba.shape), 'model', proba=proba, want_graph_labels=True, debug=False)
        edges, _, _ = sess.graph()
        assert len(edges) > 3 * K + 64
        want = segment_graph_cut_general(labels, proba, None, None, 1., 'model')
        assert np.array_equal(res['graph_labels'], want)
        assert np.array_equal(res['segm'], np.asarray(want)[labels])
    finally:
        sess.close()


def _volume_session(seed=3, shape=(12, 40, 48), sp=8):
    from pyimsegm_amd import superpixels as S
    rng = np.random.default_rng(seed)
    vol = ellipsoid_volume(shape).astype(np.float64) + 0.1 * rng.standard_normal(shape)
    sess = S._open_volume(vol)
    S._run_slic3d(sess, sp, 0.2, (2, 1, 1))
    return sess, rng


_GRAPH_KEYS = ('edges', 'centres', 'edge_weights', 'edge_weights_int', 'unary_int', 'graph_labels', 'segm', 'energy')


def test_fused_volume_call_from_the_neighbour_table(monkeypatch):
    """round 6: beyond 46 000 labels (BASELINE configs[4]: 298 116) the fused call builds its arcs from the symmetric neighbour table
    -- 64 slots per label, rows sorted on the device, reverse arcs by binary search (terms.hip k_tab_sort_rows / k_tab_emit) --
    instead of the mirrored K x K bitmap.  Forced here on a small volume: edges in (b, a) order, centres, integer terms, the cut and
    the class map equal the bitmap path's, and the graph call's (superpixels.py:180-242)."""
    from pyimsegm_amd import graph_cuts as G
    sess, rng = _volume_session()
    try:
        proba = rng.dirichlet(np.ones(3), size=sess.n_labels)
        pairwise = G.compute_pairwise_cost(0.3, proba.shape)
        by_bitmap = sess.segment(pairwise, 'model', proba=proba, debug=True, pinned=False)
        monkeypatch.setenv('IMSEGM_ADJACENCY_TABLE', '1')
        by_table = sess.segment(pairwise, 'model', proba=proba, debug=True, pinned=False)
        for key in _GRAPH_KEYS:
            assert np.array_equal(by_bitmap[key], by_table[key]), key
        edges, centres, _ = sess.graph()                          # (imsegm_volume_graph: its own read-out of the table)
        assert np.array_equal(by_table['edges'], edges) and np.array_equal(by_table['centres'], centres)
        assert len(edges) > 3 * sess.n_labels                      # a 6-connected supervoxel graph, not a planar one
    finally:
        sess.close()


def test_fused_volume_call_when_a_label_has_more_neighbours_than_a_table_row(monkeypatch):
    """a slab that touches 150 labels does not fit the 64 slots of a row: status IMSEGM_E_FUSED_PATH (HipFusedPathError), and
    the staged calls -- imsegm_volume_graph widens its rows -- give the segmentation"""
    from pyimsegm_amd import _hip
    from pyimsegm_amd.graph_cuts import compute_pairwise_cost, segment_graph_cut_general
    monkeypatch.setenv('IMSEGM_ADJACENCY_TABLE', '1')
    wide = (np.arange(6 * 30 * 40).reshape(6, 30, 40) // 8) % 150
    wide[3] = 150
    rng = np.random.default_rng(5)
    proba = rng.dirichlet(np.ones(2), size=151)
    sess = _hip.Volume3D(*wide.shape).set_labels(wide.astype(np.int64))
    try:
        with pytest.raises(_hip.HipFusedPathError, match='64 neighbours'):
            sess.segment(compute_pairwise_cost(0.5, proba.shape), 'model', proba=proba, pinned=False)
        with pytest.raises(_hip.HipFusedPathError):                # the prepared graph reports it at the call that reads its head
            sess.graph_prepare()
            sess.segment(compute_pairwise_cost(0.5, proba.shape), 'model', proba=proba, pinned=False)
        labels = segment_graph_cut_general(wide, proba, None, None, 0.5, 'model', _session=sess)
        monkeypatch.delenv('IMSEGM_ADJACENCY_TABLE')
        by_bitmap = sess.segment(compute_pairwise_cost(0.5, proba.shape), 'model', proba=proba, want_graph_labels=True, pinned=False)
        assert np.array_equal(by_bitmap['graph_labels'], labels)
    finally:
        sess.close()


@pytest.mark.parametrize('table', [False, True], ids=['bitmap', 'table'])
def test_graph_prepared_ahead_of_the_fused_call(monkeypatch, table):
    """imsegm_image2d_graph_prepare: the graph of the label map enqueued ahead (the volume pipeline builds it under the host's
    mixture fit) -- the fused call on it returns what the fused call alone returns; a prepared graph does not survive a new
    label map"""
    from pyimsegm_amd import graph_cuts as G
    from pyimsegm_amd import superpixels as S
    if table:
        monkeypatch.setenv('IMSEGM_ADJACENCY_TABLE', '1')
    sess, rng = _volume_session(seed=11)
    try:
        proba = rng.dirichlet(np.ones(3), size=sess.n_labels)
        pairwise = G.compute_pairwise_cost(0.3, proba.shape)
        alone = sess.segment(pairwise, 'model', proba=proba, debug=True, pinned=False)
        sess.graph_prepare()
        ahead = sess.segment(pairwise, 'model', proba=proba, debug=True, pinned=False)
        for key in _GRAPH_KEYS:
            assert np.array_equal(alone[key], ahead[key]), key
        # a new label map between the preparation and the call: the call builds the graph of the NEW map
        sess.graph_prepare()
        S._run_slic3d(sess, 6, 0.2, (2, 1, 1))
        proba2 = rng.dirichlet(np.ones(3), size=sess.n_labels)
        fresh = sess.segment(pairwise, 'model', proba=proba2, debug=True, pinned=False)
        edges, centres, _ = sess.graph()
        assert np.array_equal(fresh['edges'], edges) and np.array_equal(fresh['centres'], centres)
    finally:
        sess.close()


def test_graph_prepared_ahead_for_a_colour_image():
    """the same on a 2-D session (bitmap store, centres with two coordinates)"""
    from pyimsegm_amd import _hip
    from pyimsegm_amd import graph_cuts as G
    from pyimsegm_amd.utilities.synthetic import voronoi_image
    image = voronoi_image(120, 160, seed=4)
    sess = _hip.Image2D(120, 160).upload(image)
    try:
        sess.slic(60, 10.)
        rng = np.random.default_rng(2)
        proba = rng.dirichlet(np.ones(3), size=sess.n_labels)
        pairwise = G.compute_pairwise_cost(1., proba.shape)
        alone = sess.segment(pairwise, 'model', proba=proba, debug=True)
        sess.graph_prepare()
        ahead = sess.segment(pairwise, 'model', proba=proba, debug=True)
        for key in _GRAPH_KEYS:
            assert np.array_equal(alone[key], ahead[key]), key
    finally:
        sess.close()
