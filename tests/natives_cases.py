"""Shared inputs / known answers for the tests of the remaining natives (label histograms, ray features, cut_grid_graph)."""
import numpy as np

#: reference doctest, imsegm/region_growing.py:187-200 (object_segmentation_graphcut_pixels): two known label images
GRID_SEGM = np.array([[0] * 10, [1] * 5 + [0] * 5, [1] * 4 + [0] * 6, [0] * 6 + [1] * 4, [0] * 5 + [1] * 5, [0] * 10])
GRID_CENTRES = [(1, 2), (4, 8)]
GRID_EXPECT_SHAPE = np.array([[0, 0, 0, 0, 0, 0, 0, 0, 0, 0],
                              [2, 2, 1, 2, 2, 0, 0, 0, 0, 0],
                              [2, 2, 2, 2, 0, 0, 0, 0, 0, 0],
                              [0, 0, 0, 0, 0, 0, 2, 2, 2, 2],
                              [0, 0, 0, 0, 0, 2, 2, 2, 2, 2],
                              [0, 0, 0, 0, 0, 0, 0, 0, 0, 0]], dtype=np.int32)
GRID_EXPECT_SEED = np.array([[0, 0, 0, 0, 0, 0, 0, 0, 0, 0],
                             [1, 1, 1, 1, 1, 0, 0, 0, 0, 0],
                             [1, 1, 1, 1, 0, 0, 0, 0, 0, 0],
                             [0, 0, 0, 0, 0, 0, 2, 2, 2, 2],
                             [0, 0, 0, 0, 0, 2, 2, 2, 2, 2],
                             [0, 0, 0, 0, 0, 0, 0, 0, 0, 0]], dtype=np.int32)


def grid_problem(gc_regul, seed_size, coef_shape, labels_fg_prob=(0.1, 0.9), shape_mean_std=(50., 10.)):
    """the unary / pairwise / edge costs the reference's object_segmentation_graphcut_pixels (region_growing.py:158-258)
    hands to gco.cut_grid_graph for GRID_SEGM / GRID_CENTRES (its arithmetic, restated for the test)"""
    from scipy import stats
    segm, centres = GRID_SEGM, [np.round(c).astype(int) for c in GRID_CENTRES]
    height, width = segm.shape
    fg = np.array(labels_fg_prob)
    bg = 1. - fg
    proba = np.ones((height, width, len(centres) + 1))
    proba[:, :, 0] = bg[segm]
    for i in range(len(centres)):
        proba[:, :, i + 1] = fg[segm]
    shape = np.ones((height, width, len(centres) + 1))
    if coef_shape > 0:
        mean, std = shape_mean_std
        shape[:, :, 0] = bg[segm]
        grid_y, grid_x = np.meshgrid(range(width), range(height))
        for i, centre in enumerate(centres):
            dist = np.sqrt((grid_x - centre[0])**2 + (grid_y - centre[1])**2)
            cum = 1. - stats.norm.cdf(range(int(np.max(dist) + 1)), mean, std) + 1e-9
            shape[:, :, i + 1] = cum[dist.astype(int)]
    unary = -np.log(proba) - coef_shape * np.log(shape)
    disk = np.array([[0, 1, 0], [1, 1, 1], [0, 1, 0]], dtype=bool)          # skimage.morphology.disk(1)
    for i, pos in enumerate(centres):
        if seed_size > 0:
            assert seed_size == 1
            mask = np.zeros(segm.shape, dtype=bool)
            mask[pos[0] - 1:pos[0] + 2, pos[1] - 1:pos[1] + 2] = disk
            unary[np.logical_and(mask, segm > 0), i + 1] = 0
        else:
            unary[pos[0], pos[1], i + 1] = 0
    pairwise = (1 - np.eye(proba.shape[-1])) * gc_regul
    return unary, pairwise, np.ones((height - 1, width)), np.ones((height, width - 1))


def disc_segmentation():
    """`seg` of the ray-feature doctests (descriptors.py:1640-1653): ones with a disc of radius 40 at (50, 75) cleared"""
    seg = np.ones((100, 150), dtype=bool)
    rr, cc = np.mgrid[:100, :150]
    seg[(rr - 50)**2 + (cc - 75)**2 < 40**2] = False
    return seg


#: descriptors.py:1646-1653
RAY_DOCTESTS = [((50, 75), 45, [40, 41, 40, 41, 40, 41, 40, 41]),
                ((60, 40), 30, [74, 55, 28, 10, 5, 4, 4, 5, 9, 30, 57, 75]),
                ((40, 60), 20, [54, 57, 58, 55, 50, 43, 38, 31, 26, 24, 22, 22, 23, 26, 29, 34, 41, 48])]


def hist_segmentation():
    segm = np.zeros((10, 10), dtype=int)
    segm[1:9, 2:8] = 1
    segm[3:7, 4:6] = 2
    return segm
