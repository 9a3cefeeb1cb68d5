"""Helper processes for the numpy stages (pyimsegm_amd/hostpool.py): same numbers as the in-process functions,
usable from several threads, errors reported without killing the helper.  No GPU involved."""
import threading

import numpy as np
import pytest


@pytest.fixture(scope='module')
def case():
    from pyimsegm_amd import graph_cuts as G
    rng = np.random.default_rng(0)
    K, E = 700, 2000
    fts = rng.random((K, 9)) * np.array([50, 60, 70, 5, 5, 5, 3000, 4000, 5000])
    np.random.seed(0)
    model = G.estim_class_model(fts, 3, 'GMM', None, True)
    edges = np.stack([rng.integers(0, K - 3, E), np.zeros(E, int)], 1)
    edges[:, 1] = edges[:, 0] + 1 + rng.integers(0, 2, E)
    centres = rng.random((K, 2)) * 900
    return model, fts, edges.astype(np.int32), centres


def test_pool_matches_in_process_terms(case):
    from pyimsegm_amd.hostpool import HostMathPool, graph_cut_terms
    model, fts, edges, centres = case
    with HostMathPool(2) as pool:
        pool.set_model(model)
        for edge_type in ('model', 'model_l1', 'spatial', 'features', 'const'):
            ref = graph_cut_terms(model, fts, edges, centres, 2.0, edge_type)
            out = pool.terms(fts, edges, centres, 2.0, edge_type)
            assert all(np.array_equal(a, b) for a, b in zip(ref, out)), edge_type
        # a failing request is reported and the helper keeps serving
        with pytest.raises(RuntimeError, match='host helper failed'):
            pool.terms(fts[:, :4], edges, centres, 2.0, 'model')
        assert np.array_equal(pool.terms(fts, edges, centres, 2.0, 'model')[0], ref_proba(model, fts))
        # several threads share the helpers
        results = [None] * 6

        def work(i):
            results[i] = pool.terms(fts * (1 + i), edges, centres, 1.0 + i, 'model')

        threads = [threading.Thread(target=work, args=(i, )) for i in range(6)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        for i, out in enumerate(results):
            ref = graph_cut_terms(model, fts * (1 + i), edges, centres, 1.0 + i, 'model')
            assert all(np.array_equal(a, b) for a, b in zip(ref, out))


def ref_proba(model, fts):
    from pyimsegm_amd import graph_cuts as G
    return G.predict_proba(model, fts)


def test_edge_weights_split_is_the_same_function(case):
    """compute_edge_weights == graph extraction + edge_weights_from_graph (what the helpers evaluate)"""
    from pyimsegm_amd import graph_cuts as G
    model, fts, edges, centres = case
    proba = G.predict_proba(model, fts)
    w = G.edge_weights_from_graph(edges, centres, fts, proba, 'model')
    dist = G.compute_spatial_dist(centres, edges, relative=True)
    expect = np.clip(G.compute_edge_model(edges, proba, 'lT') / dist, 1e-3, 1e3)
    assert np.array_equal(w, expect)
    with pytest.raises(ValueError):
        G.edge_weights_from_graph(edges, centres, fts, None, 'model')
    with pytest.raises(RuntimeError):
        G.edge_weights_from_graph(edges, centres, None, proba, 'features')
