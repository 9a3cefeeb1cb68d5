"""Pin the oracle's adjacency-graph / centre restatement against the reference doctests
(/root/reference/imsegm/superpixels.py:163-168, 186-193, 211-215; graph_cuts.py:594-595)."""
import numpy as np


def test_graph2d_doctest(oracle):
    grid = np.array([[0] * 5 + [1] * 5, [2] * 5 + [3] * 5])
    v, edges = oracle.adjacency(grid)
    assert v.tolist() == [0, 1, 2, 3]
    assert edges == [[0, 1], [0, 2], [1, 3], [2, 3]]


def test_graph3d_doctest(oracle):
    grid_2d = np.array([[0] * 5 + [1] * 5, [2] * 5 + [3] * 5])
    grid = np.array([grid_2d, grid_2d + 4])
    v, edges = oracle.adjacency(grid)
    assert v.tolist() == list(range(8))
    assert edges == [[0, 1], [0, 2], [1, 3], [2, 3], [0, 4], [1, 5], [4, 5], [2, 6], [4, 6], [3, 7], [5, 7], [6, 7]]


def test_graph_missing_labels_doctest(oracle):
    # graph_cuts.py:587-595 -- labels {0,1,2,4,5,6}: label 3 is absent
    segments = np.array([[0] * 3 + [1] * 5 + [2] * 4, [4] * 4 + [5] * 5 + [6] * 3])
    v, edges = oracle.adjacency(segments)
    assert v.tolist() == [0, 1, 2, 4, 5, 6]
    assert edges == [[0, 1], [1, 2], [0, 4], [1, 4], [1, 5], [2, 5], [4, 5], [2, 6], [5, 6]]


def test_centers_doctest(oracle):
    segm = np.array([[0] * 6 + [1] * 5, [0] * 6 + [2] * 5])
    assert oracle.centers(segm).tolist() == [[0.5, 2.5], [0.0, 8.0], [1.0, 8.0]]
    assert oracle.centers(np.array([segm, segm, segm])).tolist() == [[1.0, 0.5, 2.5], [1.0, 0.0, 8.0], [1.0, 1.0, 8.0]]


def test_centers_missing_label(oracle):
    # superpixels.py:218 -- labels without pixels -> [-1] * ndim
    segm = np.array([[0, 0, 2, 2]])
    assert oracle.centers(segm).tolist() == [[0.0, 0.5], [-1.0, -1.0], [0.0, 2.5]]


def test_graph_against_numpy_restatement(oracle):
    """same result as the numpy formulation of superpixels.py:115-177 on a random label map"""
    rng = np.random.default_rng(0)
    grid = rng.integers(0, 23, (40, 57)) * 3
    vertices = np.unique(grid)
    rev = {v: i for i, v in enumerate(vertices)}
    dense = np.vectorize(rev.get)(grid)
    down = np.c_[dense[:-1, :].ravel(), dense[1:, :].ravel()]
    right = np.c_[dense[:, :-1].ravel(), dense[:, 1:].ravel()]
    all_edges = np.vstack([right, down])
    all_edges = all_edges[all_edges[:, 0] != all_edges[:, 1], :]
    all_edges = np.sort(all_edges, axis=1)
    h = np.unique(all_edges[:, 0] + len(vertices) * all_edges[:, 1])
    expect = [[vertices[int(e % len(vertices))], vertices[int(e / len(vertices))]] for e in h]
    v, edges = oracle.adjacency(grid)
    assert v.tolist() == vertices.tolist()
    assert edges == [[int(a), int(b)] for a, b in expect]


def test_label_cc_matches_scipy(oracle):
    """oracle restatement of skimage.measure.label (superpixels.py:111) against scipy.ndimage.label run
    per value with the full 26-neighbourhood, renumbered by first occurrence in raster order"""
    from scipy import ndimage as ndi
    rng = np.random.default_rng(0)
    for shape in [(6, 9, 11), (1, 20, 30), (4, 4, 4)]:
        lab = rng.integers(0, 4, shape)
        out = oracle.label_cc(lab)
        comp = np.zeros(shape, dtype=np.int64)
        off = 0
        for v in np.unique(lab):
            if v == 0:
                continue
            cc, n = ndi.label(lab == v, ndi.generate_binary_structure(3, 3))
            comp[cc > 0] = cc[cc > 0] + off
            off += n
        first = {}
        for c in comp.ravel():
            if c and c not in first:
                first[c] = len(first) + 1
        ref = np.vectorize(lambda c: first.get(c, 0))(comp)
        assert np.array_equal(out, ref)
        assert np.all((out == 0) == (lab == 0))


def test_oracle_slic3d_properties(oracle):
    """no skimage here to pin the 3D label map: check the structural guarantees of the restatement"""
    rng = np.random.default_rng(1)
    vol = rng.random((8, 40, 44))
    seg = oracle.segment_slic_img3d_gray(vol, 9, 0.2, (3, 1, 1))
    assert seg.shape == vol.shape and seg.min() == 0
    assert len(np.unique(seg)) == seg.max() + 1
    # measure.label output is a fixed point of measure.label
    assert np.array_equal(oracle.label_cc(seg), seg)
