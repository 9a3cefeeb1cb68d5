"""Independent check of the alpha-expansion results with networkx min-cuts (no code shared with the oracle or the
HIP kernel): the labelling the oracle returns for the graph-cut terms of the reference run (tests/golden/reference.npz)
must be a fixed point of OPTIMAL expansion moves -- for every label alpha the best alpha-expansion of the final
labelling, solved exactly as an s-t min-cut, does not lower the integer energy -- and must beat the trivial labellings.
(gco-wrapper itself exists nowhere in the container; its known-answer doctests are pinned in test_oracle_graphcut.py.)"""
import os

import numpy as np
import pytest

nx = pytest.importorskip('networkx')
VEC = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'reference.npz'), allow_pickle=False)


def integer_terms(weights, unary, pairwise):
    """the float -> integer conversion of pyGCO's cut_general_graph (oracle: orc_cut_general_graph)"""
    dwf = max(np.abs(weights).max() * pairwise.max(), np.abs(unary).max()) + 1e-10 if len(weights) else np.abs(unary).max() + 1e-10
    return ((weights / dwf) * 1000).astype(np.int64), ((unary / dwf) * 100000).astype(np.int64), (pairwise * 100).astype(np.int64)


def energy(labels, edges, w, u, s):
    return int(u[np.arange(len(labels)), labels].sum() + (w * s[labels[edges[:, 0]], labels[edges[:, 1]]]).sum())


def best_expansion_energy(labels, alpha, edges, w, u, s):
    """exact minimum of the energy over all alpha-expansions of `labels` (Kolmogorov-Zabih graph, networkx min-cut)"""
    n = len(labels)
    c0 = u[np.arange(n), labels].astype(np.int64)          # cost of keeping the label   (x = 0, source side)
    c1 = u[:, alpha].astype(np.int64)                      # cost of switching to alpha  (x = 1, sink side)
    const = 0
    graph = nx.DiGraph()
    for (i, j), wij in zip(edges, w):
        a = wij * s[labels[i], labels[j]]
        b = wij * s[labels[i], alpha]
        c = wij * s[alpha, labels[j]]
        d = wij * s[alpha, alpha]
        k = b + c - a - d
        assert k >= 0, 'pairwise term is not submodular'
        const += a
        c1[i] += c - a
        c1[j] += d - c
        if k > 0:
            cap = graph.get_edge_data(int(i), int(j), {'capacity': 0})['capacity']
            graph.add_edge(int(i), int(j), capacity=cap + int(k))
    for i in range(n):
        m = min(c0[i], c1[i])
        const += m
        graph.add_edge('s', i, capacity=int(c1[i] - m))     # cut when i is on the sink side: pays the switch
        graph.add_edge(i, 't', capacity=int(c0[i] - m))     # cut when i stays
    return int(const + nx.minimum_cut_value(graph, 's', 't'))


@pytest.mark.parametrize('name', ['disc', 'voronoi', 'voronoi_spatial', 'vol_u8'])
def test_oracle_labelling_is_a_fixed_point_of_optimal_expansions(oracle, name):
    edges = VEC[name + '_gc_edges'] if name + '_gc_edges' in VEC.files else VEC[name + '_edges_graph']
    weights, unary, pairwise = VEC[name + '_gc_edge_weights'], VEC[name + '_gc_unary'], VEC[name + '_gc_pairwise']
    labels = oracle.cut_general_graph(edges, weights, unary, pairwise, n_iter=-1)
    assert np.array_equal(labels, VEC[name + '_graph_labels'])
    w, u, s = integer_terms(weights, unary, pairwise)
    final = energy(labels, edges, w, u, s)
    for alpha in range(unary.shape[1]):
        assert best_expansion_energy(labels, alpha, edges, w, u, s) == final, 'an expansion of label %d still improves' % alpha
    # sanity of the construction itself: from the all-zero start the best expansion is no worse than keeping, and the
    # final energy is no worse than any constant labelling or the unary argmin
    start = np.zeros(len(labels), dtype=np.int64)
    assert best_expansion_energy(start, 1, edges, w, u, s) <= energy(start, edges, w, u, s)
    for const_label in range(unary.shape[1]):
        assert final <= energy(np.full(len(labels), const_label), edges, w, u, s)
    assert final <= energy(np.argmin(u, axis=1), edges, w, u, s)
