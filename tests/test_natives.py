"""CPU side of the remaining natives: the reference's own compiled features_cython (oracle/_ref) on the reference's doctest
vectors, and gco.cut_grid_graph's known answers (region_growing.py:187-200) through the oracle's alpha-expansion."""
import numpy as np
import pytest

import natives_cases as NC


def test_cut_grid_graph_known_answers(oracle):
    for (gc_regul, seed, coef), expect in (((0., 0, 0.5), NC.GRID_EXPECT_SHAPE), ((.5, 1, 0.), NC.GRID_EXPECT_SEED)):
        unary, pairwise, cost_v, cost_h = NC.grid_problem(gc_regul, seed, coef)
        labels = oracle.cut_grid_graph(unary, pairwise, cost_v, cost_h, n_iter=999)
        assert np.array_equal(np.asarray(labels).reshape(NC.GRID_SEGM.shape), expect)


def test_reference_natives_on_their_doctests(ref_cython):
    if ref_cython is None:
        pytest.skip('oracle/_ref not built')
    segm = NC.hist_segmentation()
    hist = np.array(ref_cython.computeLabelHistogram2d(np.array(segm[2:5, 4:7], dtype=np.int16), np.ones((3, 3), dtype=np.int16), 3))
    assert hist.tolist() == [0, 5, 4]                                          # descriptors.py:1480-1481
    seg = NC.disc_segmentation()
    for position, step, expect in NC.RAY_DOCTESTS:
        rays = np.array(ref_cython.computeRayFeaturesBinary2d(np.array(seg, dtype=np.int8), np.array(position, dtype=np.int32),
                                                              float(step), 1))
        assert rays.astype(int).tolist() == expect


def test_host_helpers():
    from pyimsegm_amd import descriptors as D
    assert D.adjust_bounding_box_crop((50, 50), (15, 15), (20, 45)) == ((13, 38), (28, 50), (0, 0), (15, 12))
    assert D.adjust_bounding_box_crop((50, 50), (80, 80), (20, 20)) == ((0, 0), (50, 50), (20, 20), (70, 70))
    dirs = D._ray_directions(45.)
    assert dirs.shape == (8, 2) and dirs.dtype == np.float32 and np.allclose(np.abs(dirs).max(axis=1), 1.)
