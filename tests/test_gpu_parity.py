"""GPU parity tests: every HIP stage (called through the C ABI) against the CPU oracle on the same
seeded inputs.  Integer outputs (label maps, edges, graph-cut labels) must be bit-exact; float
descriptors within 1e-5 of the reference semantics (in fact ~1e-12 / bit-exact for uint8 images)."""
import numpy as np
import pytest

from pyimsegm_amd.utilities.synthetic import disc_image, voronoi_image

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def hip():
    from pyimsegm_amd import _hip
    _hip.default_context()
    return _hip


def _params(img, sp_size, regul):
    n_seg = int(np.prod(img.shape[:2]) / (sp_size**2))
    compact = (sp_size * regul)**1.5
    return n_seg, compact


CASES = [
    ('disc256', lambda: disc_image(256), 18, 0.2),
    ('vor512', lambda: voronoi_image(512, 512), 23, 0.2),         # has an oversize component
    ('vor_ragged', lambda: voronoi_image(301, 517, seed=3), 15, 0.3),
    ('float_img', lambda: np.random.default_rng(0).random((125, 150, 3)), 20, 0.2),
    # BASELINE configs[3]: drosophila_ovary_slice-sized image with the driver's defaults (slic_size=35)
    ('ovary_size', lambda: voronoi_image(647, 1024, seed=100), 35, 0.2),
]


@pytest.mark.parametrize('name,make,sp,regul', CASES, ids=[c[0] for c in CASES])
def test_slic_bit_exact(hip, oracle, name, make, sp, regul):
    img = make()
    ref_labels, info = oracle.segment_slic_img2d(img, sp, regul, return_internals=True)
    n_seg, compact = _params(img, sp, regul)
    im = hip.Image2D(*img.shape[:2]).upload(img)
    k = im.slic(n_seg, compact, sigma=1., normalize=2)
    lab = im.get_lab()
    assert np.array_equal(lab, info['pre'].reshape(lab.shape)), 'pre-processed Lab planes differ'
    nearest = im.get_nearest()
    assert np.array_equal(nearest, info['nearest'][0]), 'k-means assignment differs'
    labels = im.get_labels()
    assert labels.dtype == np.int64
    assert np.array_equal(labels, ref_labels), 'connectivity-enforced label map differs'
    assert k == ref_labels.max() + 1


SLICO_CASES = [
    ('voronoi', lambda: voronoi_image(150, 210, seed=5), 14, 0.2),
    ('disc', lambda: disc_image(256), 18, 0.3),
    ('float_noise', lambda: np.random.default_rng(3).random((97, 131, 3)), 11, 0.1),
    ('ovary_size', lambda: voronoi_image(647, 1024, seed=101), 35, 0.2),
]


@pytest.mark.parametrize('name,make,sp,regul', SLICO_CASES, ids=[c[0] for c in SLICO_CASES])
def test_slico_bit_exact(hip, oracle, name, make, sp, regul):
    """slic_zero=True (segment_slic_img2d(slico=True), superpixels.py:63): raw assignment and final map"""
    from pyimsegm_amd import superpixels
    img = make()
    ref_labels, info = oracle.segment_slic_img2d(img, sp, regul, return_internals=True, slico=True)
    plain = oracle.segment_slic_img2d(img, sp, regul)
    assert not np.array_equal(plain, ref_labels), 'SLICO must differ from plain SLIC on this input'
    n_seg, compact = _params(img, sp, regul)
    im = hip.Image2D(*img.shape[:2]).upload(img)
    k = im.slic(n_seg, compact, sigma=1., normalize=2, slic_zero=True)
    assert np.array_equal(im.get_nearest(), info['nearest'][0]), 'k-means assignment differs'
    assert np.array_equal(im.get_labels(), ref_labels)
    assert k == ref_labels.max() + 1
    # the candidate-overflow path (whole table scan) divides by the same maxima
    im.slic(n_seg, compact, sigma=1., normalize=2, slic_zero=True, max_candidates=3)
    assert np.array_equal(im.get_labels(), ref_labels)
    # the same session afterwards gives plain SLIC again
    im.slic(n_seg, compact, sigma=1., normalize=2)
    assert np.array_equal(im.get_labels(), plain)
    im.close()
    assert np.array_equal(superpixels.segment_slic_img2d(img, sp, regul, slico=True), ref_labels)


def test_slic_candidate_overflow_path(hip, oracle):
    """force the kernel's global-memory fallback (more candidates than LDS slots)"""
    img = disc_image(256)
    ref_labels = oracle.segment_slic_img2d(img, 18, 0.2)
    n_seg, compact = _params(img, 18, 0.2)
    im = hip.Image2D(256, 256).upload(img)
    im.slic(n_seg, compact, sigma=1., normalize=2, max_candidates=3)
    assert np.array_equal(im.get_labels(), ref_labels)


def test_slic_start_label_and_no_blur(hip, oracle):
    img = voronoi_image(200, 240, seed=9)
    n_seg, compact = _params(img, 16, 0.25)
    ref = oracle.slic(img, n_seg, compact, sigma=0., normalize=(float(img.min()), float(img.max())), start_label=1)
    im = hip.Image2D(200, 240).upload(img)
    im.slic(n_seg, compact, sigma=0., normalize=1, start_label=1)
    assert np.array_equal(im.get_labels(), ref)


@pytest.mark.parametrize('dtype', ['u8', 'f32', 'f64'])
def test_color_stats(hip, oracle, dtype):
    rng = np.random.default_rng(11)
    h, w, k = 150, 203, 37
    if dtype == 'u8':
        img = rng.integers(0, 256, (h, w, 3)).astype(np.uint8)
    elif dtype == 'f32':
        img = rng.random((h, w, 3)).astype(np.float32)
    else:
        img = rng.random((h, w, 3)) * 3 - 1
    seg = (rng.integers(0, k, (h // 10 + 1, w // 10 + 1)).repeat(10, 0).repeat(10, 1)[:h, :w] * 2).astype(np.int32)
    im = hip.Image2D(h, w).upload(img).set_labels(seg)
    mean, energy, var = im.color_stats()
    img32 = np.asarray(img, dtype=np.float32)
    mean_ref = oracle.color2d_mean(img32, seg)
    energy_ref = oracle.color2d_energy(img32, seg)
    var_ref = oracle.color2d_variance(img32, seg, mean_ref.astype(np.float32))
    if dtype == 'u8':
        assert np.array_equal(mean, mean_ref) and np.array_equal(energy, energy_ref)
    assert np.allclose(mean, mean_ref, rtol=1e-12, atol=1e-14)
    assert np.allclose(energy, energy_ref, rtol=1e-12, atol=1e-14)
    assert np.allclose(var, var_ref, rtol=1e-10, atol=1e-12)
    assert np.all(mean[1::2] == 0)     # odd labels are empty -> stay 0 (features_cython.pyx:76)


def test_graph_and_centres(hip, oracle):
    img = voronoi_image(301, 517, seed=3)
    labels = oracle.segment_slic_img2d(img, 15, 0.3)
    im = hip.Image2D(301, 517).upload(img).set_labels(labels)
    edges, centres, present = im.graph()
    v_ref, e_ref = oracle.adjacency(labels)
    assert edges.tolist() == e_ref
    assert np.flatnonzero(present).tolist() == v_ref.tolist()
    assert np.array_equal(centres, oracle.centers(labels))
    # label map with holes in the id range
    sparse = (labels * 3).astype(np.int32)
    im.set_labels(sparse)
    edges, centres, present = im.graph()
    v_ref, e_ref = oracle.adjacency(sparse)
    assert edges.tolist() == e_ref and np.flatnonzero(present).tolist() == v_ref.tolist()
    assert np.array_equal(centres, oracle.centers(sparse))


def test_graphcut_doctest_vector(hip):
    """graph_cuts.py:700-703 through the C ABI"""
    np.random.seed(0)
    segments = np.array([[0] * 3 + [2] * 3 + [4] * 3 + [6] * 3 + [8] * 3, [1] * 3 + [3] * 3 + [5] * 3 + [7] * 3 + [9] * 3])
    proba = np.array([[0.1] * 6 + [0.9] * 4, [0.9] * 6 + [0.1] * 4], dtype=float).T
    proba += (0.5 - np.random.random(proba.shape)) * 0.2
    p = np.clip(proba, 0.01, 0.99)
    unary = np.abs(-np.log(p))
    im = hip.Image2D(*segments.shape).set_labels(segments)
    edges, centres, _ = im.graph()
    d = np.sqrt(((centres[edges[:, 0]] - centres[edges[:, 1]])**2).sum(axis=1))
    weights = np.clip(1. / (d / d.mean()), 1e-3, 1e3)
    labels = hip.cut_general_graph(edges, weights, unary, 1. - np.eye(2))
    assert labels.dtype == np.int32
    assert np.array_equal(labels[segments], np.array([[1] * 9 + [0] * 6] * 2))


@pytest.mark.parametrize('seed,K,C', [(0, 40, 2), (1, 300, 3), (2, 2000, 3), (3, 500, 5), (4, 64, 8)])
def test_graphcut_random_graphs(hip, oracle, seed, K, C):
    rng = np.random.default_rng(seed)
    side = int(np.ceil(np.sqrt(K)))
    ids = np.arange(side * side).reshape(side, side)
    pairs = np.vstack([np.c_[ids[:, :-1].ravel(), ids[:, 1:].ravel()], np.c_[ids[:-1, :].ravel(), ids[1:, :].ravel()]])
    pairs = pairs[(pairs < K).all(axis=1)]
    extra = rng.integers(0, K, (K // 3, 2))
    extra = extra[extra[:, 0] != extra[:, 1]]
    pairs = np.unique(np.sort(np.vstack([pairs, extra]), axis=1), axis=0).astype(np.int32)
    weights = np.clip(rng.lognormal(0, 1, len(pairs)), 1e-3, 1e3)
    proba = rng.dirichlet(np.ones(C) * 0.6, K)
    unary = np.abs(-np.log(np.clip(proba, 0.01, 0.99)))
    pairwise = 2.0 * (1 - np.eye(C))
    ref, e_ref = oracle.cut_general_graph(pairs, weights, unary, pairwise, return_energy=True)
    out, e = hip.cut_general_graph(pairs, weights, unary, pairwise, return_energy=True)
    assert e == e_ref
    assert np.array_equal(out, ref)


@pytest.mark.parametrize('seed,K,C', [(7, 400, 3), (8, 900, 4)])
def test_graphcut_pairwise_that_is_not_a_metric(hip, oracle, seed, K, C):
    """cut_general_graph takes any symmetric pairwise matrix (/root/reference/imsegm/graph_cuts.py:735-744).  The kernel's
    shortcut for a move that repeats the last accepted label is only taken when the integer matrix is a metric;
    here V(a, a) = 1 != 0 -- not a metric, every move term still submodular, so the result is defined -- and every move is run"""
    rng = np.random.default_rng(seed)
    side = int(np.ceil(np.sqrt(K)))
    ids = np.arange(side * side).reshape(side, side)
    pairs = np.vstack([np.c_[ids[:, :-1].ravel(), ids[:, 1:].ravel()], np.c_[ids[:-1, :].ravel(), ids[1:, :].ravel()]])
    pairs = pairs[(pairs < K).all(axis=1)].astype(np.int32)
    weights = np.clip(rng.lognormal(0, 1, len(pairs)), 1e-3, 1e3)
    proba = rng.dirichlet(np.ones(C) * 0.6, K)
    unary = np.abs(-np.log(np.clip(proba, 0.01, 0.99)))
    pairwise = 1. + 2. * (1 - np.eye(C))
    ref, e_ref = oracle.cut_general_graph(pairs, weights, unary, pairwise, return_energy=True)
    out, e = hip.cut_general_graph(pairs, weights, unary, pairwise, return_energy=True)
    assert e == e_ref
    assert np.array_equal(out, ref)


@pytest.mark.parametrize('level,regs', [(0, 0), (1, 0), (2, 0), (3, 0), (4, 0), (2, 1), (3, 1), (4, 1)])
def test_graphcut_every_lds_placement(hip, oracle, level, regs, monkeypatch):
    """the kernel is compiled once per placement of its arrays (graphcut.hip, template parameter LVL: scratch in global memory ...
    everything in LDS); all of them against the oracle on a graph that takes several relabellings and push rounds per move"""
    monkeypatch.setenv('IMSEGM_GC_LDS_LEVEL', str(level))
    if not regs:
        monkeypatch.setenv('IMSEGM_GC_NO_TOPO_REGS', '1')          # (regs: the arcs of a node in registers, K <= 1024)
    rng = np.random.default_rng(11)
    side, C = 26, 4
    K = side * side
    ids = np.arange(K).reshape(side, side)
    pairs = np.vstack([np.c_[ids[:, :-1].ravel(), ids[:, 1:].ravel()], np.c_[ids[:-1, :].ravel(), ids[1:, :].ravel()],
                       np.c_[ids[:-1, :-1].ravel(), ids[1:, 1:].ravel()]])
    hub = np.c_[np.full(14, K // 2 + 3), rng.choice(K // 2, 14, replace=False)]         # a node with more arcs than registers hold
    pairs = np.unique(np.sort(np.vstack([pairs, hub]), axis=1), axis=0).astype(np.int32)
    weights = np.clip(rng.lognormal(0, 1, len(pairs)), 1e-3, 1e3)
    blobs = (np.hypot(*(np.indices((side, side)) - side / 2.)) // 4).astype(int).ravel() % C          # rings: label changes far from seeds
    proba = np.full((K, C), 0.15)
    proba[np.arange(K), blobs] = 0.55
    proba *= rng.uniform(0.5, 1.5, proba.shape)
    proba /= proba.sum(axis=1, keepdims=True)
    unary = np.abs(-np.log(np.clip(proba, 0.01, 0.99)))
    pairwise = 1.5 * (1 - np.eye(C))
    ref, e_ref = oracle.cut_general_graph(pairs, weights, unary, pairwise, return_energy=True)
    out, e = hip.cut_general_graph(pairs, weights, unary, pairwise, return_energy=True)
    assert e == e_ref and np.array_equal(out, ref)
    assert len(np.unique(ref)) > 1


def _large_graph(seed, side, C, extra_share=0.1):
    """a side x side grid graph with a few long-range edges, smooth class regions with noisy probabilities: moves that take many
    relabelling levels and push rounds"""
    rng = np.random.default_rng(seed)
    K = side * side
    ids = np.arange(K).reshape(side, side)
    pairs = np.vstack([np.c_[ids[:, :-1].ravel(), ids[:, 1:].ravel()], np.c_[ids[:-1, :].ravel(), ids[1:, :].ravel()]])
    extra = rng.integers(0, K, (int(K * extra_share), 2))
    pairs = np.unique(np.sort(np.vstack([pairs, extra[extra[:, 0] != extra[:, 1]]]), axis=1), axis=0).astype(np.int32)
    weights = np.clip(rng.lognormal(0, 1, len(pairs)), 1e-3, 1e3)
    rows, cols = np.indices((side, side))
    region = ((np.sin(rows / 17.) + np.cos(cols / 23.) + 2.) * C / 4.).astype(int).ravel() % C
    proba = np.full((K, C), 0.2)
    proba[np.arange(K), region] = 0.5
    proba *= rng.uniform(0.4, 1.6, proba.shape)
    proba /= proba.sum(axis=1, keepdims=True)
    return pairs, weights, np.abs(-np.log(np.clip(proba, 0.01, 0.99)))


@pytest.mark.parametrize('seed,side,C,blocks', [(21, 100, 3, 0), (22, 150, 4, 0), (23, 120, 3, 3), (24, 96, 5, 1)])
def test_graphcut_by_the_whole_device_against_the_oracle(hip, oracle, seed, side, C, blocks, monkeypatch):
    """graphs beyond the LDS of one CU (9 216 .. 22 500 sites) go to the cooperative grid-wide kernel (graphcut.hip
    k_alpha_expansion_grid): labelling and energy of the oracle's alpha-expansion (Dinic max-flow), with one workgroup per CU,
    with three workgroups, with a single one"""
    if blocks:
        monkeypatch.setenv('IMSEGM_GC_GRID_BLOCKS', str(blocks))
    pairs, weights, unary = _large_graph(seed, side, C)
    pairwise = 1.2 * (1 - np.eye(C))
    ref, e_ref = oracle.cut_general_graph(pairs, weights, unary, pairwise, return_energy=True)
    out, e = hip.cut_general_graph(pairs, weights, unary, pairwise, return_energy=True)
    assert e == e_ref and np.array_equal(out, ref)
    assert len(np.unique(ref)) == C


@pytest.mark.parametrize('n_iter,C', [(1, 3), (3, 4), (-1, 8), (2, 8)])
def test_graphcut_by_the_whole_device_with_a_fixed_number_of_cycles(hip, oracle, n_iter, C):
    """gco's n_iter > 0 (that many cycles over the labels, stopping early when a cycle changes nothing) and eight classes on
    10 000 sites: the grid-wide kernel follows the oracle through both schedules"""
    pairs, weights, unary = _large_graph(40 + C, 100, C)
    pairwise = 0.9 * (1 - np.eye(C))
    ref, e_ref = oracle.cut_general_graph(pairs, weights, unary, pairwise, n_iter=n_iter, return_energy=True)
    out, e = hip.cut_general_graph(pairs, weights, unary, pairwise, n_iter=n_iter, return_energy=True)
    assert e == e_ref and np.array_equal(out, ref)


@pytest.mark.parametrize('side,n_iter,C', [(24, 1, 3), (30, 4, 4), (100, 2, 3), (100, 5, 4)])
def test_graphcut_fixed_cycles_with_a_pairwise_matrix_that_is_not_a_metric(hip, oracle, side, n_iter, C):
    """with V(a, a) != 0 (not a metric: GCO's moves are not exact optima) the move that repeats the last accepted
    label must be run in the fixed-cycle schedule too (n_iter > 0) -- by the single workgroup (576 / 900 sites) and by the
    grid-wide kernel (10 000 sites); the oracle runs every move"""
    pairs, weights, unary = _large_graph(60 + side + C, side, C)
    pairwise = 0.7 + 1.1 * (1 - np.eye(C))
    ref, e_ref = oracle.cut_general_graph(pairs, weights, unary, pairwise, n_iter=n_iter, return_energy=True)
    out, e = hip.cut_general_graph(pairs, weights, unary, pairwise, n_iter=n_iter, return_energy=True)
    assert e == e_ref and np.array_equal(out, ref)


@pytest.mark.timeout(120)
def test_graphcut_by_the_whole_device_when_a_workgroup_never_arrives(hip, oracle, monkeypatch):
    """on a GPU shared with another process a workgroup of the grid-wide kernel may not become resident; every wait of
    its barrier is bounded (2 s), the launch is given up as a whole and the single workgroup cuts the graph from scratch -- here
    one workgroup leaves at once (IMSEGM_GC_GRID_TEST_ABSENT): same labelling and energy as the oracle, one fall-back counted"""
    pairs, weights, unary = _large_graph(71, 100, 3)
    pairwise = 1.2 * (1 - np.eye(3))
    ref, e_ref = oracle.cut_general_graph(pairs, weights, unary, pairwise, return_energy=True)
    before = hip.gc_grid_fallbacks()
    out, e = hip.cut_general_graph(pairs, weights, unary, pairwise, return_energy=True)
    assert hip.gc_grid_fallbacks() == before and e == e_ref and np.array_equal(out, ref)
    monkeypatch.setenv('IMSEGM_GC_GRID_TEST_ABSENT', '1')
    out, e = hip.cut_general_graph(pairs, weights, unary, pairwise, return_energy=True)
    assert hip.gc_grid_fallbacks() == before + 1
    assert e == e_ref and np.array_equal(out, ref)


@pytest.mark.timeout(300)
def test_graphcut_by_the_whole_device_from_several_threads(hip):
    """three worker threads (a context and a HIP stream each) cut graphs of 14 400 sites at the same time: cooperative launches are
    taken one at a time (two grids that each hold a part of the CUs would wait for each other for ever); every result equals the
    one of its graph cut alone"""
    from concurrent.futures import ThreadPoolExecutor
    graphs = [_large_graph(50 + i, 120, 3) for i in range(6)]
    pairwise = 1.1 * (1 - np.eye(3))
    alone = [hip.cut_general_graph(p, w, u, pairwise, return_energy=True) for p, w, u in graphs]
    with ThreadPoolExecutor(max_workers=3) as pool:
        together = list(pool.map(lambda g: hip.cut_general_graph(g[0], g[1], g[2], pairwise, return_energy=True), graphs * 2))
    for i, (labels, energy) in enumerate(together):
        assert energy == alone[i % 6][1] and np.array_equal(labels, alone[i % 6][0])


def test_graphcut_by_the_whole_device_equals_one_workgroup(hip, monkeypatch):
    """160 000 sites, 335 000 edges: the grid-wide kernel against the single workgroup working out of global memory -- and with
    a pairwise matrix that is not a metric (no move skipped)"""
    pairs, weights, unary = _large_graph(31, 400, 3, extra_share=0.05)
    for pairwise in (0.8 * (1 - np.eye(3)), 0.5 + 0.8 * (1 - np.eye(3))):
        out, e = hip.cut_general_graph(pairs, weights, unary, pairwise, return_energy=True)
        monkeypatch.setenv('IMSEGM_GC_ONE_WORKGROUP', '1')
        single, e_single = hip.cut_general_graph(pairs, weights, unary, pairwise, return_energy=True)
        monkeypatch.delenv('IMSEGM_GC_ONE_WORKGROUP')
        assert e == e_single and np.array_equal(out, single)
        assert len(np.unique(out)) == 3


def test_graphcut_no_edges_and_errors(hip):
    unary = np.array([[3., 1., 2.], [0.5, 0.5, 0.1], [1., 1., 1.]])
    out = hip.cut_general_graph(np.zeros((0, 2), dtype=np.int32), np.zeros(0), unary, 1 - np.eye(3))
    assert out.tolist() == [1, 2, 0]
    with pytest.raises(hip.HipError):
        hip.cut_general_graph(np.array([[1, 0]]), np.ones(1), unary, 1 - np.eye(3))
    with pytest.raises(hip.HipError):
        hip.cut_general_graph(np.array([[0, 1]]), np.ones(1), unary, np.array([[0, 1, 2], [0, 0, 1], [2, 1, 0.]]))


def test_gathers(hip, oracle):
    rng = np.random.default_rng(5)
    labels = rng.integers(0, 50, (97, 131)).astype(np.int32)
    gl = rng.integers(0, 3, 50).astype(np.int32)
    proba = rng.random((50, 3))
    im = hip.Image2D(97, 131).set_labels(labels)
    segm, soft = im.gather(gl, proba)
    assert np.array_equal(segm, gl[labels]) and segm.dtype == np.int32
    assert np.array_equal(soft, proba[labels])


def test_full_size_properties(hip):
    """BASELINE config 2 size (2048 x 2048): size-independent properties of the label map"""
    from scipy import ndimage as ndi
    img = voronoi_image(2048, 2048)
    n_seg, compact = _params(img, 46, 0.2)
    im = hip.Image2D(2048, 2048).upload(img)
    k = im.slic(n_seg, compact)
    labels = im.get_labels()
    assert labels.min() == 0 and labels.max() == k - 1
    counts = np.bincount(labels.ravel(), minlength=k)
    min_size = int(0.5 * 2048 * 2048 / 2025)
    assert counts.min() >= min_size                      # every kept superpixel is at least min_size
    # every superpixel is 4-connected: components of equal-label regions == number of labels
    same_r = labels[:, 1:] == labels[:, :-1]
    same_d = labels[1:, :] == labels[:-1, :]
    n = labels.size
    idx = np.arange(n).reshape(labels.shape)
    import scipy.sparse as sp
    import scipy.sparse.csgraph as csg
    rows = np.concatenate([idx[:, :-1][same_r], idx[:-1, :][same_d]])
    cols = np.concatenate([idx[:, 1:][same_r], idx[1:, :][same_d]])
    ncomp, _ = csg.connected_components(sp.coo_matrix((np.ones(len(rows), dtype=np.int8), (rows, cols)), shape=(n, n)),
                                        directed=False)
    assert ncomp == k
    # idempotence of the deterministic pipeline: a second run gives the identical map
    im2 = hip.Image2D(2048, 2048).upload(img)
    im2.slic(n_seg, compact)
    assert np.array_equal(im2.get_labels(), labels)
    # descriptors: mean of per-superpixel means weighted by size == global mean (linearity)
    mean, energy, var = im.color_stats()
    glob = img.reshape(-1, 3).astype(np.float64).mean(axis=0)
    assert np.allclose((mean * counts[:, None]).sum(axis=0) / n, glob, rtol=1e-12)


def _flat_dot(h, w):
    img = np.full((h, w, 3), 128, dtype=np.uint8)
    img[h // 3, w // 4] = 129
    return img


def _two_halves(h, w):
    img = np.zeros((h, w, 3), dtype=np.uint8)
    img[:, w // 2:] = (200, 50, 120)
    return img


EDGE_CASES = [
    ('ovary_slice_size', lambda: voronoi_image(647, 1024, seed=100), 35, 0.2),    # BASELINE config 4 image shape
    ('tiny_20x30', lambda: voronoi_image(20, 30, seed=5, nb_seeds=4), 6, 0.3),     # smaller than one tile
    ('one_superpixel', lambda: voronoi_image(40, 50, seed=6, nb_seeds=3), 40, 0.2),  # K = 1
    ('thin_strip', lambda: voronoi_image(9, 400, seed=8, nb_seeds=6), 8, 0.25),
    ('low_compactness', lambda: voronoi_image(150, 170, seed=9), 12, 0.02),         # large colour weight
    ('binary_0_1', lambda: (voronoi_image(96, 96, seed=2) > 100).astype(np.uint8), 10, 0.2),   # min 0, max 1: no scaling
    ('flat_with_one_dot', lambda: _flat_dot(96, 128), 12, 0.2),      # exact ties everywhere: the near-tie path decides
    ('two_flat_halves', lambda: _two_halves(120, 90), 15, 0.2),
    ('one_row', lambda: voronoi_image(1, 300, seed=3, nb_seeds=5), 6, 0.3),
    ('one_column', lambda: voronoi_image(300, 1, seed=3, nb_seeds=5), 6, 0.3),
    ('four_by_four', lambda: voronoi_image(4, 4, seed=3, nb_seeds=5), 2, 0.3),
    ('tile_plus_one_column', lambda: voronoi_image(32, 65, seed=3, nb_seeds=5), 9, 0.3),      # 64 x 32 tile geometry
    ('tile_plus_one_row', lambda: voronoi_image(33, 64, seed=3, nb_seeds=5), 9, 0.3),
]


@pytest.mark.parametrize('name,make,sp,regul', EDGE_CASES, ids=[c[0] for c in EDGE_CASES])
def test_slic_edge_cases_bit_exact(hip, oracle, name, make, sp, regul):
    img = make()
    ref_labels = oracle.segment_slic_img2d(img, sp, regul)
    n_seg, compact = _params(img, sp, regul)
    im = hip.Image2D(*img.shape[:2]).upload(img)
    k = im.slic(n_seg, compact, sigma=1., normalize=2)
    assert np.array_equal(im.get_labels(), ref_labels)
    assert k == ref_labels.max() + 1
    # the rest of the stage chain on the same label map
    edges, centres, present = im.graph()
    v_ref, e_ref = oracle.adjacency(ref_labels)
    assert edges.tolist() == e_ref and np.flatnonzero(present).tolist() == v_ref.tolist()
    assert np.array_equal(centres, oracle.centers(ref_labels))
    mean, energy, var = im.color_stats()
    img32 = img.astype(np.float32)
    assert np.array_equal(mean, oracle.color2d_mean(img32, ref_labels))
    assert np.array_equal(energy, oracle.color2d_energy(img32, ref_labels))


def test_slic_exact_path_matches_fast_path(hip, oracle):
    """normalize=0 disables the fp32 pre-selection (exact fp64 loop for every pixel): same labels"""
    img = voronoi_image(200, 230, seed=12).astype(np.float64) / 255.
    n_seg, compact = _params(img, 14, 0.2)
    ref = oracle.slic(img, n_seg, compact, sigma=1.)
    im = hip.Image2D(200, 230).upload(img)
    im.slic(n_seg, compact, sigma=1., normalize=0)
    exact = im.get_labels()
    im.slic(n_seg, compact, sigma=1., normalize=2)      # data spans [0, 1] only approximately -> scaled
    assert np.array_equal(exact, ref)


def test_slic_randomised_sweep(hip, oracle):
    """seeded sweep over image kinds (piecewise, uint8 noise, float noise, smooth), sizes, superpixel sizes and
    compactness from 0.09 to 400: the fp32 pre-selection with its margin, the sorted break and the exact
    near-tie path must reproduce the oracle bit for bit everywhere"""
    rng = np.random.default_rng(2024)
    for case in range(16):
        H, W = int(rng.integers(40, 300)), int(rng.integers(40, 400))
        kind = case % 4
        if kind == 0:
            img = voronoi_image(H, W, seed=int(rng.integers(1 << 30)))
        elif kind == 1:
            img = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
        elif kind == 2:
            img = rng.random((H, W, 3))
        else:
            yy, xx = np.mgrid[:H, :W]
            img = np.stack([np.sin(yy / 7.0) * 0.5 + 0.5, np.cos(xx / 11.0) * 0.5 + 0.5, (yy + xx) / (H + W)], axis=-1)
        sp = int(rng.integers(5, 40))
        regul = float(rng.choice([0.02, 0.1, 0.2, 0.5, 1.0, 3.0]))
        n_seg, compact = _params(img, sp, regul)
        if n_seg < 1:
            continue
        ref = oracle.segment_slic_img2d(img, sp, regul)
        im = hip.Image2D(H, W).upload(img)
        im.slic(n_seg, compact, sigma=1., normalize=2)
        got = im.get_labels()
        im.close()
        assert np.array_equal(got, ref), 'case %d kind %d %dx%d sp %d regul %g: %d px differ' % (
            case, kind, H, W, sp, regul, np.count_nonzero(got != ref))


def test_slic_sweeps_replayed_from_a_hip_graph(hip, oracle, monkeypatch):
    """IMSEGM_SLIC_GRAPH=1: capture on the first image of a session, replay on the next (same parameters), re-capture
    when they change -- label maps as without the graph"""
    from pyimsegm_amd.utilities.synthetic import voronoi_image
    monkeypatch.setenv('IMSEGM_SLIC_GRAPH', '1')
    sess = hip.Image2D(200, 264)
    try:
        for seed, sp in ((5, 20), (6, 20), (7, 26)):
            img = voronoi_image(200, 264, seed=seed)
            n_seg, compact = _params(img, sp, 0.2)
            sess.upload(img)
            sess.slic(n_seg, compact)
            assert np.array_equal(sess.get_labels(), oracle.segment_slic_img2d(img, sp, 0.2))
    finally:
        sess.close()
