"""Output side of the driver (/root/reference/experiments_segmentation/run_segm_slic_model_graphcut.py:350,
366-375): the `slic_mean` debug image against the real scikit-image's `label2rgb(kind='avg')`, `assume_bg_on_boundary` on the
device against the reference's numpy statements, the narrow result formats against the default ones."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('name', ['voronoi', 'float'])
def test_slic_mean_equals_label2rgb_avg_of_scikit_image(name):
    """imsegm/pipelines.py:93 -- debug_visual['slic_mean'] = label2rgb(slic, image, kind='avg'); label 0 included (bg_label = -1 in 0.18)"""
    sys.path.insert(0, GOLDEN)
    from make_golden_skimage import CASES_2D, make_input
    from pyimsegm_amd import pipelines as pipe
    ref, sk = np.load(os.path.join(GOLDEN, 'label2rgb.npz')), np.load(os.path.join(GOLDEN, 'skimage.npz'))
    expr, sp_size, regul = CASES_2D[name]
    image = make_input(expr)
    res = pipe._ResidentImage(image, {'color': ['mean']}, sp_size, regul)
    try:
        assert np.array_equal(res.slic, sk[name + '_final'])
        got = res.mean_colour_image()
    finally:
        res.close()
    want = ref[name + '_avg']
    # uint8 images: the sums are exact, the means equal scikit-image's to the last bit or two; float images pass through the
    # statistic kernel in float32, as the reference's Cython descriptors read them (descriptors.py:233): 1e-7
    tol = 1e-12 * np.abs(want).max() if image.dtype == np.uint8 else 1e-7
    assert got.shape == want.shape and np.abs(got - want).max() <= tol
    assert np.abs(got[res.slic == 0]).max() > 0            # superpixel 0 is no background


def _reference_assume_bg(segm, bg_label, boundary_size):
    """the statements of labeling.py:743-752 and data_io.py:1024-1027 (2-D), restated for the comparison"""
    size = int(boundary_size)
    bg_pixels = np.hstack([segm[:size, :], segm[:, :size].T, segm[-size:, :], segm[:, -size:].T])
    boundary_lb = int(np.argmax(np.bincount(bg_pixels.ravel())))
    used = np.unique(segm)
    lut = list(range(max(int(used.max()), bg_label) + 1))
    lut[boundary_lb] = bg_label
    lut[bg_label] = boundary_lb
    return np.array(lut)[segm]


def test_assume_bg_on_boundary_doctests_and_random_maps():
    from pyimsegm_amd.labeling import assume_bg_on_boundary
    segm = np.zeros((6, 12), dtype=int)
    segm[1:4, 4:] = 2
    want = np.zeros((6, 12), dtype=int)
    want[1:4, 4:] = 2
    assert np.array_equal(assume_bg_on_boundary(segm, boundary_size=1), want)            # labeling.py:727-734
    segm[segm == 0] = 1
    assert np.array_equal(assume_bg_on_boundary(segm, boundary_size=1), want)            # labeling.py:735-742
    rng = np.random.default_rng(5)
    for shape, nb, size, bg, dtype in [((647, 1024), 3, 105, 0, np.int32), ((200, 300), 5, 1, 2, np.int64), ((64, 48), 4, 40, 0, np.uint8),
                                       ((90, 70), 6, 70, 1, np.int32), ((33, 129), 9000, 3, 0, np.int32)]:
        segm = rng.integers(0, nb, shape).astype(dtype)
        segm[:, : shape[1] // 3] = nb - 1                                               # one label owns the left third
        got = assume_bg_on_boundary(segm.copy(), bg_label=bg, boundary_size=size)
        assert got.dtype == np.int64 and np.array_equal(got, _reference_assume_bg(segm.astype(np.int64), bg, size)), (shape, nb, size)
    with pytest.raises(Exception):
        assume_bg_on_boundary(np.full((8, 8), -1, dtype=np.int32))
    # the reference's own function, lifted from its file (tests/golden/make_golden.py -> labeling.npz)
    g = np.load(os.path.join(GOLDEN, 'labeling.npz'))
    for name in ('a', 'b'):
        for size in (1, 3):
            out = assume_bg_on_boundary(g['segm_' + name].copy(), bg_label=0, boundary_size=size)
            assert np.array_equal(out, g['bg_%s_%d' % (name, size)]), (name, size)


def test_narrow_result_formats_equal_the_default_ones():
    from pyimsegm_amd import pipelines as pipe
    from pyimsegm_amd.descriptors import FEATURES_SET_COLOR
    from pyimsegm_amd.graph_cuts import estim_class_model
    from pyimsegm_amd.utilities.synthetic import voronoi_image
    image = voronoi_image(333, 517, seed=11)
    np.random.seed(0)
    _, features = pipe.compute_color2d_superpixels_features(image, FEATURES_SET_COLOR, 20, 0.2)
    model = estim_class_model(features, 3, 'GMM', None, True)
    segm, soft = pipe.segment_color2d_slic_features_model_graphcut(image, model, FEATURES_SET_COLOR, 20, 0.2, 2., 'model')
    segm8, soft32 = pipe.segment_color2d_slic_features_model_graphcut(image, model, FEATURES_SET_COLOR, 20, 0.2, 2., 'model',
                                                                     segm_dtype=np.uint8, soft_dtype=np.float32)
    assert segm.dtype == np.int32 and soft.dtype == np.float64                          # the reference's dtypes by default
    assert segm8.dtype == np.uint8 and np.array_equal(segm8, segm.astype(np.uint8))
    assert soft32.dtype == np.float32 and np.array_equal(soft32, soft.astype(np.float32))
    segm_only, none = pipe.segment_color2d_slic_features_model_graphcut(image, model, FEATURES_SET_COLOR, 20, 0.2, 2., 'model',
                                                                        segm_dtype=np.uint8, soft_dtype=False)
    assert none is None and np.array_equal(segm_only, segm8)
