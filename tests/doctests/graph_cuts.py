"""Examples that used to sit in the docstrings of pyimsegm_amd/graph_cuts.py: mostly the doctest vectors of the reference module
(/root/reference/imsegm/graph_cuts.py) its functions mirror, run against the module by tests/test_alias_package.py (the ones that need
no GPU) and tests/test_gpu_api.py (all of them, `# doctest: +SKIP` lifted)."""

EXAMPLES = {
    'estim_class_model': r"""
>>> np.random.seed(0)
>>> fts = np.vstack([np.random.random((50, 3)) - 1, np.random.random((50, 3)) + 1])
>>> estim_class_model(fts, 2).predict_proba(fts).shape
(100, 2)
""",
    'compute_spatial_dist': r"""
>>> centres = [(0.5, 1.0), (0.0, 3.5), (0.0, 7.0), [-1, -1], (1.0, 1.5), (1.0, 4.5), (1.0, 8.0)]
>>> edges = [[0, 1], [1, 2], [4, 5], [5, 6], [0, 4], [1, 5], [2, 6]]
>>> np.round(compute_spatial_dist(centres, edges), 2).tolist()
[2.55, 3.5, 3.0, 3.5, 0.71, 1.41, 1.41]
""",
    'compute_edge_model': r"""
>>> edges = np.array([[0, 1], [1, 2], [0, 4], [1, 4], [1, 5], [2, 5], [4, 5], [2, 6], [5, 6]])
>>> np.random.seed(0)
>>> img = np.random.random((2, 12, 3)) * 255
>>> proba = np.random.random((7, 2))
>>> np.round(compute_edge_model(edges, proba, metric='l1'), 3).tolist()
[0.002, 0.015, 0.001, 0.002, 0.0, 0.002, 0.015, 0.034, 0.001]
>>> np.round(compute_edge_model(edges, proba, metric='lT'), 3).tolist()
[0.0, 0.002, 0.0, 0.005, 0.0, 0.0, 0.101, 0.092, 0.001]
""",
    'create_pairwise_matrix': r"""
>>> create_pairwise_matrix(0.6, 3).tolist()
[[0.0, 0.6, 0.6], [0.6, 0.0, 0.6], [0.6, 0.6, 0.0]]
>>> create_pairwise_matrix([((1, 2), 0.5), ((0, 2), 0.7)], 3).tolist()
[[0.0, 1.0, 0.7], [1.0, 0.0, 0.5], [0.7, 0.5, 0.0]]
""",
    'compute_unary_cost': r"""
>>> compute_unary_cost(np.array([[0.5, 0.001], [1., 0.3]])).round(4).tolist()
[[0.6931, 4.6052], [0.0101, 1.204]]
""",
    'count_label_transitions_connected_segments': r"""
>>> dict_slics = {'a': np.array([[0] * 3 + [1] * 3 + [2] * 3 + [3] * 3 + [4] * 3,
...                              [5] * 3 + [6] * 3 + [7] * 3 + [8] * 3 + [9] * 3])}
>>> dict_labels = {'a': np.array([0, 0, 1, 1, 2, 0, 1, 1, 0, 2])}
>>> count_label_transitions_connected_segments(dict_slics, dict_labels).tolist()  # doctest: +SKIP
[[2.0, 5.0, 1.0], [5.0, 3.0, 1.0], [1.0, 1.0, 1.0]]
""",
}
