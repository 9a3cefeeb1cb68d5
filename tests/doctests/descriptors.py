"""Examples that used to sit in the docstrings of pyimsegm_amd/descriptors.py: mostly the doctest vectors of the reference module
(/root/reference/imsegm/descriptors.py) its functions mirror, run against the module by tests/test_alias_package.py (the ones that need
no GPU) and tests/test_gpu_api.py (all of them, `# doctest: +SKIP` lifted)."""

EXAMPLES = {
    '_same_plane': r"""
>>> _same_plane(np.zeros((125, 150, 3)), np.zeros((150, 125)))  # doctest: +ELLIPSIS
Traceback (most recent call last):
...
pyimsegm_amd.utilities.ImageDimensionError: ndarrays - image and segmentation do not match (125, 150, 3) vs (150, 125)
""",
    '_same_shape': r"""
>>> _same_shape(np.zeros((125, 150)), np.zeros((150, 125)))  # doctest: +ELLIPSIS
Traceback (most recent call last):
...
pyimsegm_amd.utilities.ImageDimensionError: ndarrays - image and segmentation do not match (125, 150) vs (150, 125)
""",
    '_three_channels': r"""
>>> _three_channels(np.zeros((200, 250, 1)))  # doctest: +ELLIPSIS
Traceback (most recent call last):
...
pyimsegm_amd.utilities.ImageDimensionError: image is not RGB with dims (200, 250, 1)
""",
    '_report_unknown_groups': r"""
>>> _report_unknown_groups({'color': [], 'texture': []})
['texture']
""",
    '_report_unknown_names': r"""
>>> _report_unknown_names(['mean', 'average'])
['average']
""",
    'hip_img2d_color_mean': r"""
>>> image = np.zeros((2, 10, 3))
>>> image[:, 2:6, 0] = 1
>>> image[:, 3:7, 1] = 3
>>> image[:, 4:9, 2] = 2
>>> segm = np.array([[0, 0, 0, 0, 0, 1, 1, 1, 1, 1],
...                  [0, 0, 0, 0, 0, 1, 1, 1, 1, 1]])
>>> hip_img2d_color_mean(image, segm)  # doctest: +SKIP
array([[0.6, 1.2, 0.4],
       [0.2, 1.2, 1.6]])
""",
    '_channel_medians': r"""
>>> image = np.zeros((2, 10, 3))
>>> image[:, 2:6, 0] = 1
>>> image[:, 3:8, 1] = 3
>>> image[:, 4:9, 2] = 2
>>> segm = np.array([[0, 0, 0, 0, 1, 1, 1, 1, 1, 1],
...                  [0, 0, 0, 0, 1, 1, 1, 1, 1, 1]])
>>> _channel_medians(image, segm).tolist()
[[0.5, 0.0, 0.0], [0.0, 3.0, 2.0]]
""",
    'hip_img3d_gray_mean': r"""
>>> image = np.zeros((2, 3, 8))
>>> image[0, :, 2:6] = 1
>>> image[1, :, 3:7] = 3
>>> segm = np.array([[[0, 0, 0, 0, 1, 1, 1, 1]] * 3,
...                  [[2, 2, 2, 2, 3, 3, 3, 3]] * 3])
>>> hip_img3d_gray_mean(image, segm).tolist()  # doctest: +SKIP
[0.5, 0.5, 0.75, 2.25]
""",
    'hip_img3d_gray_energy': r"""
>>> image = np.zeros((2, 3, 8))
>>> image[0, :, 2:6] = 1
>>> image[1, :, 3:7] = 3
>>> segm = np.array([[[0, 0, 0, 0, 1, 1, 1, 1]] * 3,
...                  [[2, 2, 2, 2, 3, 3, 3, 3]] * 3])
>>> hip_img3d_gray_energy(image, segm).tolist()  # doctest: +SKIP
[0.5, 0.5, 2.25, 6.75]
""",
    'hip_img3d_gray_std': r"""
>>> image = np.zeros((2, 3, 8))
>>> image[0, :, 2:6] = 1
>>> image[1, :, 3:7] = 3
>>> segm = np.array([[[0, 0, 0, 0, 1, 1, 1, 1]] * 3,
...                  [[2, 2, 2, 2, 3, 3, 3, 3]] * 3])
>>> np.round(hip_img3d_gray_std(image, segm), 4).tolist()  # doctest: +SKIP
[0.5, 0.5, 1.299, 1.299]
""",
    'compute_image3d_gray_statistic': r"""
>>> image = np.zeros((2, 3, 8))
>>> image[0, :, 2:6] = 1
>>> image[1, :, 3:7] = 3
>>> segm = np.array([[[0, 0, 0, 0, 1, 1, 1, 1]] * 3,
...                  [[2, 2, 2, 2, 5, 5, 5, 5]] * 3])
>>> features, names = compute_image3d_gray_statistic(image, segm)  # doctest: +SKIP
>>> np.round(features, 3).tolist()  # doctest: +SKIP +NORMALIZE_WHITESPACE
[[0.5, 0.5, 0.5, 0.5, 0.25], [0.5, 0.5, 0.5, 0.5, -0.25], [0.75, 1.299, 2.25, 0.0, 0.75],
 [0.0, 0.0, 0.0, 0.0, 0.0], [0.0, 0.0, 0.0, 0.0, 0.0], [2.25, 1.299, 6.75, 3.0, -1.125]]
>>> names  # doctest: +SKIP
['gray_mean', 'gray_std', 'gray_energy', 'gray_median', 'gray_meanGrad']
""",
    'compute_image2d_color_statistic': r"""
>>> image = np.zeros((2, 10, 3))
>>> image[:, 2:6, 0] = 1
>>> image[:, 3:7, 1] = 3
>>> image[:, 4:9, 2] = 2
>>> segm = np.array([[0, 0, 0, 0, 0, 1, 1, 1, 1, 1],
...                  [0, 0, 0, 0, 0, 1, 1, 1, 1, 1]])
>>> features, names = compute_image2d_color_statistic(image, segm)  # doctest: +SKIP
>>> features.shape  # doctest: +SKIP
(2, 15)
""",
    'create_filter_bank_lm_2d': r"""
>>> filters, names = create_filter_bank_lm_2d(6, SHORT_FILTERS_SIGMAS, 2)
>>> [f.shape for f in filters][:5]
[(2, 13, 13), (2, 13, 13), (1, 13, 13), (1, 13, 13), (1, 13, 13)]
>>> names[:5]
['sigma1.4-edge', 'sigma1.4-bar', 'sigma1.4-Gauss', 'sigma1.4-GaussLap', 'sigma1.4-GaussLap2']
""",
    'compute_texture_desc_lm_img2d_clr': r"""
>>> h, w, step = 30, 20, 5
>>> np.random.seed(0)
>>> seg = (np.arange(h)[:, None] // step) * (w // step) + np.arange(w)[None, :] // step
>>> img = np.random.random((h, w, 3))
>>> features, names = compute_texture_desc_lm_img2d_clr(img, seg, ['mean', 'std', 'median'],
...                                                     bank_type='short')  # doctest: +SKIP
>>> features.shape  # doctest: +SKIP
(24, 135)
""",
    'compute_selected_features_gray3d': r"""
>>> np.random.seed(0)
>>> img = np.random.random((2, 10, 15))
>>> slic = np.zeros((2, 10, 15), dtype=int)
>>> slic[:, :, :7] += 1
>>> slic[1, :, :] += 2
>>> fts, names = compute_selected_features_gray3d(img, slic, {'color': ('mean', 'std', 'median')})  # doctest: +SKIP
>>> fts.shape  # doctest: +SKIP
(4, 3)
>>> names  # doctest: +SKIP
['gray_mean', 'gray_std', 'gray_median']
""",
    'compute_selected_features_gray2d': r"""
>>> image = np.zeros((2, 10))
>>> image[0, 2:6] = 1
>>> image[1, 3:7] = 3
>>> segm = np.array([[0, 0, 0, 0, 0, 1, 1, 1, 1, 1],
...                  [0, 0, 0, 0, 0, 1, 1, 1, 1, 1]])
>>> features, names = compute_selected_features_gray2d(image, segm, {'color': ('mean', 'std', 'median')})  # doctest: +SKIP
>>> np.round(features, 3).tolist()  # doctest: +SKIP
[[0.9, 1.136, 0.5], [0.7, 1.187, 0.0]]
""",
    'compute_selected_features_color2d': r"""
>>> image = np.zeros((2, 10, 3))
>>> image[:, 2:6, 0] = 1
>>> image[:, 3:7, 1] = 3
>>> image[:, 4:9, 2] = 2
>>> segm = np.array([[0, 0, 0, 0, 0, 1, 1, 1, 1, 1],
...                  [0, 0, 0, 0, 0, 1, 1, 1, 1, 1]])
>>> features, names = compute_selected_features_color2d(image, segm,
...                                   {'color': ('mean', 'std', 'median')})  # doctest: +SKIP
>>> np.round(features, 3)  # doctest: +SKIP
array([[0.6 , 1.2 , 0.4 , 0.49, 1.47, 0.8 , 1.  , 0.  , 0.  ],
       [0.2 , 1.2 , 1.6 , 0.4 , 1.47, 0.8 , 0.  , 0.  , 2.  ]])
""",
    'adjust_bounding_box_crop': r"""
>>> adjust_bounding_box_crop((50, 50), (7, 7), (20, 20))
((17, 17), (24, 24), (0, 0), (7, 7))
>>> adjust_bounding_box_crop((50, 50), (15, 15), (20, 45))
((13, 38), (28, 50), (0, 0), (15, 12))
>>> adjust_bounding_box_crop((50, 50), (15, 15), (5, 5))
((0, 0), (13, 13), (2, 2), (15, 15))
>>> adjust_bounding_box_crop((50, 50), (80, 80), (20, 20))
((0, 0), (50, 50), (20, 20), (70, 70))
""",
    'hip_label_hist_seg2d': r"""
>>> segm = np.zeros((10, 10), dtype=int)
>>> segm[1:9, 2:8] = 1
>>> segm[3:7, 4:6] = 2
>>> hip_label_hist_seg2d(segm[2:5, 4:7], np.ones((3, 3)), 3).tolist()  # doctest: +SKIP
[0.0, 5.0, 4.0]
""",
    'compute_label_hist_segm': r"""
>>> segm = np.zeros((10, 10), dtype=int)
>>> segm[1:9, 2:8] = 1
>>> segm[3:7, 4:6] = 2
>>> hist, size = compute_label_hist_segm(segm, [6, 6], np.ones((3, 3)), 3)  # doctest: +SKIP
>>> hist.tolist(), float(size)  # doctest: +SKIP
([0.0, 7.0, 2.0], 9.0)
""",
    'hip_ray_features_seg2d': r"""
>>> seg_empty = np.zeros((100, 150), dtype=bool)
>>> hip_ray_features_seg2d(seg_empty, (50, 75), 90).tolist()  # doctest: +SKIP
[-1.0, -1.0, -1.0, -1.0]
""",
}
