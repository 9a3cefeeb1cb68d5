"""The ``imsegm`` alias package exposes the reference's module names on top of pyimsegm_amd."""


def test_alias_modules_are_the_same_objects():
    import imsegm
    import imsegm.descriptors as seg_fts
    import imsegm.graph_cuts
    import imsegm.pipelines as seg_pipe
    import imsegm.superpixels
    from imsegm.utilities import ImageDimensionError
    import pyimsegm_amd.descriptors
    import pyimsegm_amd.pipelines
    assert seg_fts is pyimsegm_amd.descriptors
    assert seg_pipe is pyimsegm_amd.pipelines
    seg_fts.USE_CYTHON = False          # what the reference driver does (run_segm_slic_model_graphcut.py:59)
    assert pyimsegm_amd.descriptors.USE_CYTHON is False
    seg_fts.USE_CYTHON = True
    assert issubclass(ImageDimensionError, TypeError)
    for name in ('pipe_color2d_slic_features_model_graphcut', 'estim_model_classes_group',
                 'segment_color2d_slic_features_model_graphcut', 'compute_color2d_superpixels_features',
                 'train_classif_color2d_slic_features', 'wrapper_compute_color2d_slic_features_labels'):
        assert callable(getattr(seg_pipe, name))
    assert imsegm.__version__


def test_host_side_doctests():
    """the examples of the host mirror that need no GPU (tests/doctests: they sat in the docstrings before)"""
    import warnings
    from tests.doctests import MODULES, run_examples
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        for key in MODULES:
            failed, attempted = run_examples(key)
            assert failed == 0 and attempted > 0, key
