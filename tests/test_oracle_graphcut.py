"""Pin the oracle's alpha-expansion restatement (gco-wrapper `cut_general_graph`) against the
reference's known-answer doctests and against brute force on tiny graphs."""
import itertools

import numpy as np
from scipy import stats


def _unary_cost(proba, min_prob=0.01):
    # graph_cuts.py:523-540
    proba = proba.copy()
    proba[proba < min_prob] = min_prob
    proba[proba > 1 - min_prob] = 1 - min_prob
    return np.abs(-np.log(proba))


def _spatial_rel(centres, edges):
    d = np.sqrt(((centres[edges[:, 0]] - centres[edges[:, 1]])**2).sum(axis=1))
    return d / d.mean()


def test_doctest_graph_cuts_700(oracle):
    """graph_cuts.py:680-703: 10 superpixels, 2 labels, gc_regul=1, edge_type='spatial'"""
    np.random.seed(0)
    segments = np.array([[0] * 3 + [2] * 3 + [4] * 3 + [6] * 3 + [8] * 3, [1] * 3 + [3] * 3 + [5] * 3 + [7] * 3 + [9] * 3])
    proba = np.array([[0.1] * 6 + [0.9] * 4, [0.9] * 6 + [0.1] * 4], dtype=float).T
    proba += (0.5 - np.random.random(proba.shape)) * 0.2
    unary = _unary_cost(proba)
    assert np.allclose(unary[4], [4.60517019, 0.0797884])   # graph_cuts.py:686-696
    _, edges = oracle.adjacency(segments)
    edges = np.array(edges, dtype=np.int32)
    weights = np.ones(len(edges)) / _spatial_rel(oracle.centers(segments), edges)
    weights = np.clip(weights, 1e-3, 1e3)
    pairwise = 1. * (np.ones(2) - np.eye(2))
    labels = oracle.cut_general_graph(edges, weights, unary, pairwise, n_iter=-1)
    assert labels.dtype == np.int32
    expect = np.array([[1] * 9 + [0] * 6, [1] * 9 + [0] * 6])
    assert np.array_equal(labels[segments], expect)


def _region_growing_case(oracle, gc_regul, coef_shape):
    """region_growing.py:42-150 restated for its two doctests (:72-75)"""
    slic = np.array([[0] * 3 + [1] * 3 + [2] * 3 + [3] * 3 + [4] * 3, [5] * 3 + [6] * 3 + [7] * 3 + [8] * 3 + [9] * 3])
    segm = np.array([[0] * 15, [1] * 12 + [0] * 3])
    centres = [np.array((1, 7))]
    labels_fg_prob = np.array((0.1, 0.9))
    hist = np.zeros((10, 2))
    for s, a in zip(slic.ravel(), segm.ravel()):
        hist[s, a] += 1
    labels = np.argmax(hist, axis=1)
    labels_bg_prob = 1. - labels_fg_prob
    slic_points = oracle.centers(slic)
    proba = np.ones((10, 2))
    proba[:, 0] = labels_bg_prob[labels]
    proba[:, 1] = labels_fg_prob[labels]
    shape = np.ones((10, 2))
    if coef_shape > 0:
        shape[:, 0] = labels_bg_prob[labels]
        dist = np.sqrt(((slic_points - centres[0])**2).sum(axis=1))
        cdf = stats.norm.cdf(range(int(np.max(dist) + 1)), 50., 10.)
        shape[:, 1] = (1. - cdf + 1e-9)[dist.astype(int)]
    _, edges = oracle.adjacency(slic)
    edges = np.array(edges)
    unary = -np.log(proba) - coef_shape * np.log(shape)
    unary[slic[1, 7], 1] = 0
    min_unary = -np.log(1 - 0.01)
    unary[unary < min_unary] = min_unary
    proba_fg = labels_fg_prob[labels]
    dist = np.abs(proba_fg[edges[:, 0]] - proba_fg[edges[:, 1]])
    weights = np.exp(-dist / (2 * np.std(dist)**2))
    weights /= _spatial_rel(slic_points, edges)
    pairwise = (1 - np.eye(2)) * gc_regul
    return oracle.cut_general_graph(edges, weights, unary, pairwise, n_iter=999)


def test_doctest_region_growing_72(oracle):
    assert _region_growing_case(oracle, 0., 1.).tolist() == [0, 0, 0, 0, 0, 1, 1, 1, 1, 0]


def test_doctest_region_growing_74(oracle):
    assert _region_growing_case(oracle, 1., 0.).tolist() == [0, 0, 0, 0, 0, 1, 1, 1, 1, 0]


def _int_energy(edges, w, unary, smooth, labels):
    e = sum(int(unary[i, l]) for i, l in enumerate(labels))
    for (a, b), ww in zip(edges, w):
        e += int(ww) * int(smooth[labels[a], labels[b]])
    return e


def test_two_label_expansion_is_global_optimum(oracle):
    """with 2 labels and a metric pairwise term one expansion sweep is exact: compare with brute force,
    including the cut convention (ties -> as many sites as possible take the expanded label)"""
    import ctypes as C
    rng = np.random.default_rng(5)
    for trial in range(30):
        K = 9
        pairs = [(a, b) for a in range(K) for b in range(a + 1, K) if rng.random() < 0.35]
        if not pairs:
            continue
        edges = np.array(pairs, dtype=np.int32)
        w = rng.integers(0, 6, len(edges)).astype(np.int32)
        unary = rng.integers(0, 12, (K, 2)).astype(np.int32)
        smooth = np.array([[0, 3], [3, 0]], dtype=np.int32)
        labels = np.empty(K, dtype=np.int32)
        lib = oracle.lib()
        e = lib.orc_alpha_expansion_int(oracle._p(edges), C.c_int(len(edges)), oracle._p(w), oracle._p(unary),
                                        C.c_int(K), C.c_int(2), oracle._p(smooth), C.c_int(-1), oracle._p(labels))
        best = min(_int_energy(edges, w, unary, smooth, lab) for lab in itertools.product((0, 1), repeat=K))
        assert e == best == _int_energy(edges, w, unary, smooth, labels)


def test_multilabel_expansion_local_optimality(oracle):
    """3 labels: the result cannot be improved by any single expansion move (checked by brute force
    over all subsets for a small graph) and the reported energy is the labelling's energy"""
    import ctypes as C
    rng = np.random.default_rng(7)
    K, nc = 8, 3
    pairs = [(a, b) for a in range(K) for b in range(a + 1, K) if rng.random() < 0.4]
    edges = np.array(pairs, dtype=np.int32)
    w = rng.integers(1, 5, len(edges)).astype(np.int32)
    unary = rng.integers(0, 20, (K, nc)).astype(np.int32)
    smooth = (4 * (1 - np.eye(nc))).astype(np.int32)
    labels = np.empty(K, dtype=np.int32)
    e = oracle.lib().orc_alpha_expansion_int(oracle._p(edges), C.c_int(len(edges)), oracle._p(w), oracle._p(unary),
                                             C.c_int(K), C.c_int(nc), oracle._p(smooth), C.c_int(-1), oracle._p(labels))
    assert e == _int_energy(edges, w, unary, smooth, labels)
    for alpha in range(nc):
        for mask in range(1 << K):
            cand = [alpha if (mask >> i) & 1 else labels[i] for i in range(K)]
            assert _int_energy(edges, w, unary, smooth, cand) >= e


def test_no_edges_is_argmin(oracle):
    unary = np.array([[3., 1., 2.], [0.5, 0.5, 0.1], [1., 1., 1.]])
    labels = oracle.cut_general_graph(np.zeros((0, 2), dtype=np.int32), np.zeros(0), unary, 1 - np.eye(3))
    assert labels.tolist() == [1, 2, 0]
