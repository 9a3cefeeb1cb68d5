"""Golden vectors produced by the reference's OWN code (tests/golden/make_golden.py lifts the functions out of
/root/reference with `ast`, and runs the reference's features_cython.pyx compiled verbatim) against
(a) the CPU oracle and (b) the host-side mirror functions.  The GPU counterparts live in test_gpu_golden.py."""
import os

import numpy as np
import pytest

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def _load(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def test_oracle_graph_matches_reference(oracle):
    g = _load('graph.npz')
    v, e = oracle.adjacency(g['seg2d'])
    assert v.tolist() == g['seg2d_vertices'].tolist()
    assert e == g['seg2d_edges'].tolist()
    v, e = oracle.adjacency(g['seg3d'])
    assert v.tolist() == g['seg3d_vertices'].tolist()
    assert e == g['seg3d_edges'].tolist()
    centres = oracle.centers(g['seg3d'])
    np.testing.assert_allclose(centres, g['seg3d_centres'], rtol=0, atol=1e-12)


def test_oracle_descriptors_match_reference_cython(oracle):
    d = _load('descriptors.npz')
    mean = oracle.color2d_mean(d['img2d'], d['seg2d'])
    np.testing.assert_allclose(mean, d['mean2d'], rtol=1e-6, atol=1e-6)          # the reference is built with -ffast-math
    np.testing.assert_allclose(oracle.color2d_energy(d['img2d'], d['seg2d']), d['energy2d'], rtol=1e-6, atol=1e-4)
    var = oracle.color2d_variance(d['img2d'], d['seg2d'], np.array(d['mean2d'], dtype=np.float32))
    np.testing.assert_allclose(var, d['var2d'], rtol=1e-6, atol=1e-4)
    np.testing.assert_allclose(oracle.gray3d_stat(d['vol'], d['segv'], 'mean'), d['meanv'], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(oracle.gray3d_stat(d['vol'], d['segv'], 'energy'), d['energyv'], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(oracle.gray3d_stat(d['vol'], d['segv'], 'var', np.array(d['meanv'], dtype=np.float32)),
                               d['varv'], rtol=1e-6, atol=1e-7)


def test_host_graph_cut_terms_match_reference():
    from pyimsegm_amd import graph_cuts as gc
    t = _load('graph_cut_terms.npz')
    edges, proba = t['edges'], t['proba']
    centres = [tuple(c) for c in t['centres']]
    for metric in ('l1', 'l2', 'lT'):
        np.testing.assert_array_equal(gc.compute_edge_model(edges, proba, metric), t['edge_model_' + metric])
    np.testing.assert_array_equal(gc.compute_spatial_dist(centres, edges, relative=False), t['spatial'])
    np.testing.assert_array_equal(gc.compute_spatial_dist(t['centres'], edges, relative=True), t['spatial_rel'])
    np.testing.assert_array_equal(gc.compute_unary_cost(proba), t['unary'])
    np.testing.assert_array_equal(gc.compute_pairwise_cost(2.0, proba.shape), t['pairwise_scalar'])
    np.testing.assert_array_equal(gc.compute_pairwise_cost([((0, 1), 0.5), ((1, 2), 3.0)], proba.shape), t['pairwise_pairs'])
    np.testing.assert_array_equal(gc.compute_pairwise_cost(t['pairwise_matrix_in'], proba.shape), t['pairwise_matrix'])


def test_host_texture_bank_matches_reference():
    from pyimsegm_amd import descriptors as d
    t = _load('texture.npz')
    bank, names = d.create_filter_bank_lm_2d(radius=8, sigmas=(np.sqrt(2), 2), nb_orient=4)
    assert list(names) == t['names'].tolist()
    for i, battery in enumerate(bank):
        np.testing.assert_allclose(np.asarray(battery), t['battery_%02d' % i], rtol=0, atol=1e-15)
        resp = d.compute_img_filter_response2d(t['gray'], battery)
        np.testing.assert_allclose(resp, t['response_%02d' % i], rtol=1e-12, atol=1e-12)
    assert 'smooth' in t.files


def test_oracle_label_histograms_match_reference(oracle):
    """labeling.py:208-280 lifted from the reference and its doctest vectors (:217-230, :259-270)"""
    g = _load('labeling.npz')
    counts = oracle.histogram_regions_labels_counts(g['slic'], g['annot'])
    assert counts.dtype == np.float64 and np.array_equal(counts, g['counts'])
    assert np.array_equal(oracle.histogram_regions_labels_norm(g['slic'], g['annot']), g['norm'])
    slic = np.array([[0] * 3 + [1] * 3 + [2] * 3] * 4 + [[4] * 3 + [5] * 3 + [6] * 3] * 4)
    segm = np.zeros(slic.shape, dtype=int)
    segm[4:, 5:] = 2
    assert oracle.histogram_regions_labels_counts(slic, segm).tolist() == \
        [[12, 0, 0], [12, 0, 0], [12, 0, 0], [0, 0, 0], [12, 0, 0], [8, 0, 4], [0, 0, 12]]
    norm = oracle.histogram_regions_labels_norm(slic, segm)
    np.testing.assert_allclose(norm[5], [2 / 3., 0, 1 / 3.], rtol=0, atol=1e-15)
    assert norm[3].tolist() == [0, 0, 0] and not np.signbit(norm).any()
    with pytest.raises(ValueError):
        oracle.histogram_regions_labels_counts(slic, segm - 1)


def test_host_boundary_background_matches_reference():
    """data_io.get_image2d_boundary_color (driver touch point) against the reference's own function; labeling.assume_bg_on_boundary
    runs on the device since round 3: the same golden vectors are checked in tests/test_gpu_output.py"""
    from pyimsegm_amd.utilities.data_io import get_image2d_boundary_color
    g = _load('labeling.npz')
    assert np.array_equal(get_image2d_boundary_color(g['colour'], size=2), g['colour_bg'])
