"""Examples that used to sit in the docstrings of pyimsegm_amd/classification.py: mostly the doctest vectors of the reference module
(/root/reference/imsegm/classification.py) its functions mirror, run against the module by tests/test_alias_package.py (the ones that need
no GPU) and tests/test_gpu_api.py (all of them, `# doctest: +SKIP` lifted)."""

EXAMPLES = {
    'create_classif_search_train_export': r"""
>>> np.random.seed(0)
>>> lbs = np.random.randint(0, 3, 150)
>>> fts = np.random.random((150, 5)) + np.tile(lbs, (5, 1)).T
>>> clf, _ = create_classif_search_train_export('DecTree', fts, lbs, nb_search_iter=0)
>>> float(np.mean(clf.predict(fts) == lbs)) > 0.9
True
>>> clf, _ = create_classif_search_train_export('KNN', fts, lbs, nb_search_iter=3, cross_val=3)
>>> clf.predict_proba(fts).shape
(150, 3)
""",
    'convert_set_features_labels_2_dataset': r"""
>>> np.random.seed(0)
>>> d_fts = {'a': np.random.random((25, 3)), 'b': np.random.random((30, 3))}
>>> d_lbs = {'a': np.random.randint(0, 2, 25), 'b': np.random.randint(0, 2, 30)}
>>> fts, lbs, sizes = convert_set_features_labels_2_dataset(d_fts, d_lbs)
>>> fts.shape, lbs.shape, sizes
((55, 3), (55,), [25, 30])
""",
    'CrossValidateGroups': r"""
>>> cv = CrossValidateGroups([2, 3, 2, 1], nb_hold_out=2)
>>> len(cv)
2
>>> [(train, test) for train, test in cv]
[([5, 6, 7], [0, 1, 2, 3, 4]), ([0, 1, 2, 3, 4], [5, 6, 7])]
""",
}
