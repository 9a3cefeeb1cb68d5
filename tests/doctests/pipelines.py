"""Examples that used to sit in the docstrings of pyimsegm_amd/pipelines.py: mostly the doctest vectors of the reference module
(/root/reference/imsegm/pipelines.py) its functions mirror, run against the module by tests/test_alias_package.py (the ones that need
no GPU) and tests/test_gpu_api.py (all of them, `# doctest: +SKIP` lifted)."""

EXAMPLES = {
    'pipe_color2d_slic_features_model_graphcut': r"""
>>> np.random.seed(0)
>>> image = np.random.random((125, 150, 3)) / 2.
>>> image[:, :75] += 0.5
>>> segm, seg_soft = pipe_color2d_slic_features_model_graphcut(image, 2, {'color': ['mean']})  # doctest: +SKIP
>>> segm.shape  # doctest: +SKIP
(125, 150)
""",
    'pipe_gray3d_slic_features_model_graphcut': r"""
>>> np.random.seed(0)
>>> image = np.random.random((5, 125, 150)) / 2.
>>> image[:, :, :75] += 0.5
>>> segm = pipe_gray3d_slic_features_model_graphcut(image, 2, {'color': ['mean']})  # doctest: +SKIP
>>> segm.shape  # doctest: +SKIP
(5, 125, 150)
""",
}
