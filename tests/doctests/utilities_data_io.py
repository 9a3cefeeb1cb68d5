"""Examples that used to sit in the docstrings of pyimsegm_amd/utilities/data_io.py: mostly the doctest vectors of the reference module
(/root/reference/imsegm/utilities/data_io.py) its functions mirror, run against the module by tests/test_alias_package.py (the ones that need
no GPU) and tests/test_gpu_api.py (all of them, `# doctest: +SKIP` lifted)."""

EXAMPLES = {
    'rgb2hsv': r"""
>>> rgb2hsv(np.array([[[1., 0., 0.], [0., 0.5, 0.5]]])).tolist()
[[[0.0, 1.0, 1.0], [0.5, 1.0, 0.5]]]
""",
    'convert_img_color_from_rgb': r"""
>>> convert_img_color_from_rgb(np.ones((50, 75, 3)), 'hsv').shape
(50, 75, 3)
""",
    'get_image2d_boundary_color': r"""
>>> img = np.zeros((5, 15), dtype=int)
>>> img[:4, 3:9] = 1
>>> int(get_image2d_boundary_color(img))
0
>>> get_image2d_boundary_color(np.ones((5, 15, 3), dtype=int), size=2).tolist()
[1, 1, 1]
>>> int(get_image2d_boundary_color(np.ones((5, 15, 3, 1), dtype=int)))
0
""",
}
