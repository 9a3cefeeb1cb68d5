"""Examples that used to sit in the docstrings of pyimsegm_amd/labeling.py: mostly the doctest vectors of the reference module
(/root/reference/imsegm/labeling.py) its functions mirror, run against the module by tests/test_alias_package.py (the ones that need
no GPU) and tests/test_gpu_api.py (all of them, `# doctest: +SKIP` lifted)."""

EXAMPLES = {
    'histogram_regions_labels_counts': r"""
>>> slic = np.array([[0] * 3 + [1] * 3 + [2] * 3] * 4 +
...                 [[4] * 3 + [5] * 3 + [6] * 3] * 4)
>>> segm = np.zeros(slic.shape, dtype=int)
>>> segm[4:, 5:] = 2
>>> histogram_regions_labels_counts(slic, segm)  # doctest: +SKIP
array([[12.,  0.,  0.],
       [12.,  0.,  0.],
       [12.,  0.,  0.],
       [ 0.,  0.,  0.],
       [12.,  0.,  0.],
       [ 8.,  0.,  4.],
       [ 0.,  0., 12.]])
""",
    'histogram_regions_labels_norm': r"""
>>> slic = np.array([[0] * 3 + [1] * 3 + [2] * 3] * 4 +
...                 [[4] * 3 + [5] * 3 + [6] * 3] * 4)
>>> segm = np.zeros(slic.shape, dtype=int)
>>> segm[4:, 5:] = 2
>>> histogram_regions_labels_norm(slic, segm)  # doctest: +SKIP
array([[1.        , 0.        , 0.        ],
       [1.        , 0.        , 0.        ],
       [1.        , 0.        , 0.        ],
       [0.        , 0.        , 0.        ],
       [1.        , 0.        , 0.        ],
       [0.66666667, 0.        , 0.33333333],
       [0.        , 0.        , 1.        ]])
""",
    'assume_bg_on_boundary': r"""
>>> segm = np.zeros((6, 12), dtype=int)
>>> segm[1:4, 4:] = 2
>>> assume_bg_on_boundary(segm, boundary_size=1)[2].tolist()  # doctest: +SKIP
[0, 0, 0, 0, 2, 2, 2, 2, 2, 2, 2, 2]
>>> segm[segm == 0] = 1
>>> out = assume_bg_on_boundary(segm, boundary_size=1)  # doctest: +SKIP
>>> out[0].tolist(), out[2].tolist()  # doctest: +SKIP
([0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0], [0, 0, 0, 0, 2, 2, 2, 2, 2, 2, 2, 2])
""",
}
