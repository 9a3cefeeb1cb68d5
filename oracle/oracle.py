"""ctypes front end of the CPU oracle (``liboracle.so``) -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import this module.  The product package ``pyimsegm_amd`` never does.

Besides the thin wrappers it restates, in numpy, the host-side glue of the reference that sits
between the native calls (parameter mapping of ``imsegm/superpixels.py:22-69``, the regular grid of
``skimage/util/_regular_grid.py``, the Gaussian taps of ``scipy.ndimage``), so that a whole
reference-equivalent ``segment_slic_img2d`` can be run on the CPU.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build():
    """compile liboracle.so (and oracle/_ref when the reference tree is present)"""
    subprocess.check_call(['make', '-s', '-C', _HERE, 'all'])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, 'liboracle.so')
        src = os.path.join(_HERE, 'imsegm_oracle.c')
        if not os.path.exists(path) or os.path.getmtime(path) < os.path.getmtime(src):
            build()
        _LIB = C.CDLL(path)
        _LIB.orc_det_cbrt.restype = C.c_double
        _LIB.orc_det_cbrt.argtypes = [C.c_double]
        _LIB.orc_det_pow24.restype = C.c_double
        _LIB.orc_det_pow24.argtypes = [C.c_double]
        _LIB.orc_adjacency.restype = C.c_long
        _LIB.orc_alpha_expansion_int.restype = C.c_int64
        _LIB.orc_cut_general_graph.restype = C.c_int64
    return _LIB


def ref_features_cython():
    """the reference's own compiled ``features_cython`` module (oracle/_ref), or None"""
    import importlib.util
    import sysconfig
    path = os.path.join(_HERE, '_ref', 'features_cython' + sysconfig.get_config_var('EXT_SUFFIX'))
    if not os.path.exists(path):
        return None
    spec = importlib.util.spec_from_file_location('features_cython', path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _p(a, t=C.c_void_p):
    return a.ctypes.data_as(t)


_dp = C.POINTER(C.c_double)

# ------------------------------------------------------------------------------------------------
# SLIC
# ------------------------------------------------------------------------------------------------


def gaussian_taps(sigma, truncate=4.0):
    """half kernel w[0..r] (w[0] = centre) of scipy.ndimage.gaussian_filter1d; None if sigma == 0"""
    if not sigma > 0:
        return None
    try:
        from scipy.ndimage import _filters
    except ImportError:                       # scipy < 1.8 (the build container's conda interpreter)
        from scipy.ndimage import filters as _filters
    r = int(truncate * float(sigma) + 0.5)
    full = _filters._gaussian_kernel1d(float(sigma), 0, r)[::-1]
    return np.ascontiguousarray(full[r:], dtype=np.float64)


def regular_grid(ar_shape, n_points):
    """skimage.util.regular_grid (util/_regular_grid.py, 0.18): (starts, steps) per axis;
    step None means ``slice(None)`` (every element)"""
    ar_shape = np.asanyarray(ar_shape)
    ndim = len(ar_shape)
    unsort_dim_idxs = np.argsort(np.argsort(ar_shape))
    sorted_dims = np.sort(ar_shape)
    space_size = float(np.prod(ar_shape))
    if space_size <= n_points:
        return [(0, None)] * ndim
    stepsizes = np.full(ndim, (space_size / n_points)**(1.0 / ndim), dtype='float64')
    if (sorted_dims < stepsizes).any():
        for dim in range(ndim):
            stepsizes[dim] = sorted_dims[dim]
            space_size = float(np.prod(sorted_dims[dim + 1:]))
            stepsizes[dim + 1:] = ((space_size / n_points)**(1.0 / (ndim - dim - 1)))
            if (sorted_dims >= stepsizes).all():
                break
    starts = (stepsizes // 2).astype(int)
    stepsizes = np.round(stepsizes).astype(int)
    slices = [(int(start), int(step)) for start, step in zip(starts, stepsizes)]
    return [slices[i] for i in unsort_dim_idxs]


def grid_centroids(shape3, n_segments):
    """_get_grid_centroids of slic_superpixels.py (0.18): K x 3 (z, y, x) float64 and steps"""
    slices = regular_grid(shape3, n_segments)
    axes = []
    for (start, step), size in zip(slices, shape3):
        axes.append(np.arange(start, size, step if step is not None else 1))
    zz, yy, xx = np.meshgrid(*axes, indexing='ij')
    centroids = np.stack([zz.ravel(), yy.ravel(), xx.ravel()], axis=-1).astype(np.float64)
    steps = [float(step) if step is not None else 1.0 for _, step in slices]
    return centroids, steps


def img_as_float64(image):
    """skimage.util.img_as_float (util/dtype.py) for the dtypes the reference feeds it, always in
    float64 (the float32-in -> float32-compute branch of skimage 0.18 is not restated)"""
    image = np.asarray(image)
    if image.dtype.kind == 'f':
        return np.ascontiguousarray(image, dtype=np.float64)
    if image.dtype.kind == 'u':
        return np.multiply(image, 1. / np.iinfo(image.dtype).max, dtype=np.float64)
    if image.dtype.kind == 'i':
        info = np.iinfo(image.dtype)
        out = np.add(image, 0.5, dtype=np.float64)
        out *= 2. / (float(info.max) - float(info.min))
        return out
    if image.dtype.kind == 'b':
        return image.astype(np.float64)
    raise ValueError('unsupported dtype %r' % image.dtype)


def slic(image, n_segments, compactness, sigma=0., spacing=None, multichannel=True, max_iter=10,
         enforce_connectivity=True, min_size_factor=0.5, max_size_factor=3, start_label=0,
         normalize=None, return_internals=False, slic_zero=False):
    """CPU restatement of ``skimage.segmentation.slic`` (0.18.x) for the two call shapes of the
    reference: H x W x 3 colour (``imsegm/superpixels.py:61``) and D x H x W gray with
    ``multichannel=False`` (``:104``).

    ``normalize``: None, or (vmin, vmax) to fold the min-max scaling of ``superpixels.py:53-54`` into
    the pre-processing (needed to keep uint8 inputs uint8 up to here).
    """
    L = lib()
    image = np.ascontiguousarray(image)
    if spacing is None:
        spacing = (1., 1., 1.)
    spacing = np.asarray(spacing, dtype=np.float64)
    sig = np.array([sigma, sigma, sigma], dtype=np.float64) / spacing
    taps = [gaussian_taps(s) for s in sig]
    tap_args = []
    for t in taps:
        if t is None:
            tap_args += [None, C.c_int(-1)]
        else:
            tap_args += [_p(t), C.c_int(len(t) - 1)]
    ratio = 1.0 / compactness
    if multichannel:
        H, W = image.shape[:2]
        D, nch = 1, 3
        assert image.ndim == 3 and image.shape[2] == 3
        if image.dtype == np.uint8:
            dtype = 0
        else:
            image = np.ascontiguousarray(image, dtype=np.float64)
            dtype = 1
        lab = np.empty((3, H, W), dtype=np.float64)
        vmin, vmax = normalize if normalize is not None else (0., 1.)
        L.orc_slic_preprocess_color2d(
            _p(image), C.c_int(dtype), C.c_int(H), C.c_int(W), C.c_int(normalize is not None),
            C.c_double(vmin), C.c_double(vmax), *tap_args, C.c_double(ratio), _p(lab))
        pre = lab
    else:
        assert image.ndim == 3 and normalize is None
        D, H, W = image.shape
        nch = 1
        img = img_as_float64(image)
        pre = np.empty((1, D, H, W), dtype=np.float64)
        L.orc_slic_preprocess_gray3d(_p(img), C.c_int(D), C.c_int(H), C.c_int(W), *tap_args,
                                     C.c_double(ratio), _p(pre))
    cent, steps = grid_centroids((D, H, W), n_segments)
    K = cent.shape[0]
    segments = np.zeros((K, 3 + nch), dtype=np.float64)
    segments[:, :3] = cent
    step = max(steps)
    # _slic_cython recomputes the integer steps from regular_grid((D, H, W), n_segments) with
    # n_segments == number of centroids passed in (segments.shape[0])
    slices_k = regular_grid((D, H, W), K)
    isteps = [int(s if s is not None else 1) for _, s in slices_k]
    nearest = np.empty((D, H, W), dtype=np.int32)
    L.orc_slic_iterate(_p(pre), C.c_int(nch), C.c_int(D), C.c_int(H), C.c_int(W), C.c_int(K),
                       _p(segments), C.c_int(isteps[0]), C.c_int(isteps[1]), C.c_int(isteps[2]),
                       C.c_double(np.float32(step)), _p(spacing), C.c_int(max_iter), C.c_int(int(bool(slic_zero))),
                       _p(nearest))
    labels = nearest + start_label
    raw = labels
    if enforce_connectivity:
        segment_size = D * H * W / K
        min_size = int(min_size_factor * segment_size)
        max_size = int(max_size_factor * segment_size)
        out = np.empty_like(labels)
        L.orc_enforce_connectivity(_p(labels), C.c_int(D), C.c_int(H), C.c_int(W),
                                   C.c_long(min_size), C.c_long(max_size), C.c_int(start_label),
                                   _p(out))
        labels = out
    labels = labels.astype(np.int64)
    if multichannel:
        labels = labels[0]
    if return_internals:
        return labels, dict(pre=pre, nearest=raw, segments=segments, K=K, steps=isteps, step=step)
    return labels


def segment_slic_img2d(img, sp_size=50, relative_compact=0.1, start_label=0, return_internals=False, slico=False):
    """imsegm/superpixels.py:22-69 on the oracle (``slico``: skimage's ``slic_zero``)"""
    img = np.asarray(img)
    nb_pixels = np.prod(img.shape[:2])
    if img.ndim == 2:
        img = np.rollaxis(np.tile(img, (3, 1, 1)), 0, 3)
    normalize = None
    if img.min() != 0. or img.max() != 1.:
        if img.dtype == np.uint8 or img.dtype == np.float64:
            normalize = (float(img.min()), float(img.max()))
        else:
            img = (img - img.min()) / float(img.max() - img.min())
    n_seg = int(nb_pixels / (sp_size**2))
    compact = (sp_size * relative_compact)**1.5
    return slic(img, n_seg, compact, sigma=1, normalize=normalize, start_label=start_label,
                return_internals=return_internals, slic_zero=slico)


def enforce_connectivity(labels, min_size, max_size, start_label=0):
    """``_enforce_label_connectivity_cython`` (scikit-image 0.18 ``_slic.pyx``) on a 2-D or 3-D int label map"""
    L = lib()
    lab = np.ascontiguousarray(labels, dtype=np.int32)
    shape3 = (1, ) + lab.shape if lab.ndim == 2 else lab.shape
    out = np.empty_like(lab)
    L.orc_enforce_connectivity(_p(lab), C.c_int(shape3[0]), C.c_int(shape3[1]), C.c_int(shape3[2]), C.c_long(int(min_size)),
                               C.c_long(int(max_size)), C.c_int(start_label), _p(out))
    return out.astype(np.int64)


def label_cc(labels):
    """skimage.measure.label(labels) restated (background 0, full connectivity, raster-order numbering)"""
    lab = np.ascontiguousarray(labels, dtype=np.int32)
    l3 = lab if lab.ndim == 3 else lab[np.newaxis]
    D, H, W = l3.shape
    out = np.empty_like(l3)
    lib().orc_label_cc(_p(l3), C.c_int(D), C.c_int(H), C.c_int(W), _p(out))
    return out.reshape(lab.shape).astype(np.int64)


def segment_slic_img3d_gray(im, sp_size=50, relative_compact=0.1, space=(1, 1, 1), start_label=0):
    """imsegm/superpixels.py:72-112 on the oracle"""
    im = np.asarray(im)
    nb_pixels = np.prod(im.shape)
    sp_vol = np.prod(sp_size / np.asarray(space, dtype=np.float32) * min(space))
    n_seg = int(nb_pixels / sp_vol)
    compact = int((sp_vol * relative_compact)**1.5)
    if im.dtype == np.float32:       # scikit-image 0.18 runs a float32 volume in float32 from end to end
        seg = slic_gray3d_float32(im, n_seg, compact, sigma=1., spacing=space, start_label=start_label)
    else:
        seg = slic(np.array(im), n_seg, compact, sigma=1, spacing=space, multichannel=False, start_label=start_label)
    return label_cc(seg)


# ------------------------------------------------------------------------------------------------
# descriptors
# ------------------------------------------------------------------------------------------------


def color2d_mean(img32, seg32):
    img32 = np.ascontiguousarray(img32, dtype=np.float32)
    seg32 = np.ascontiguousarray(seg32, dtype=np.int32)
    H, W = seg32.shape
    out = np.empty((int(seg32.max()) + 1, 3), dtype=np.float64)
    lib().orc_color2d_mean(_p(img32), _p(seg32), C.c_int(H), C.c_int(W), _p(out))
    return out


def color2d_energy(img32, seg32):
    img32 = np.ascontiguousarray(img32, dtype=np.float32)
    seg32 = np.ascontiguousarray(seg32, dtype=np.int32)
    H, W = seg32.shape
    out = np.empty((int(seg32.max()) + 1, 3), dtype=np.float64)
    lib().orc_color2d_energy(_p(img32), _p(seg32), C.c_int(H), C.c_int(W), _p(out))
    return out


def color2d_variance(img32, seg32, mean32):
    img32 = np.ascontiguousarray(img32, dtype=np.float32)
    seg32 = np.ascontiguousarray(seg32, dtype=np.int32)
    mean32 = np.ascontiguousarray(mean32, dtype=np.float32)
    H, W = seg32.shape
    out = np.empty((int(seg32.max()) + 1, 3), dtype=np.float64)
    lib().orc_color2d_variance(_p(img32), _p(seg32), C.c_int(H), C.c_int(W), _p(mean32), _p(out))
    return out


def gray3d_stat(img32, seg32, which, mean32=None):
    img32 = np.ascontiguousarray(img32, dtype=np.float32)
    seg32 = np.ascontiguousarray(seg32, dtype=np.int32)
    D, H, W = seg32.shape
    out = np.empty(int(seg32.max()) + 1, dtype=np.float64)
    m = None if mean32 is None else _p(np.ascontiguousarray(mean32, dtype=np.float32))
    lib().orc_gray3d_stat(_p(img32), _p(seg32), C.c_int(D), C.c_int(H), C.c_int(W),
                          C.c_int({'mean': 0, 'energy': 1, 'var': 2}[which]), m, _p(out))
    return out


# ------------------------------------------------------------------------------------------------
# graph
# ------------------------------------------------------------------------------------------------


def adjacency(grid):
    """(vertices ndarray, edges list[[a, b]]) as make_graph_segm_connect_grid{2d_conn4,3d_conn6}"""
    grid = np.ascontiguousarray(grid, dtype=np.int32)
    g3 = grid if grid.ndim == 3 else grid[np.newaxis]
    D, H, W = g3.shape
    nb = int(grid.max()) + 1
    vertices = np.empty(nb, dtype=np.int32)
    nv = C.c_int(0)
    cap = 64
    while True:
        edges = np.empty((cap, 2), dtype=np.int32)
        ne = lib().orc_adjacency(_p(g3), C.c_int(D), C.c_int(H), C.c_int(W), _p(vertices),
                                 C.byref(nv), _p(edges), C.c_long(cap))
        if ne >= 0:
            break
        cap *= 4
    return vertices[:nv.value].astype(np.int64), edges[:ne].tolist()


def centers(grid):
    grid = np.ascontiguousarray(grid, dtype=np.int32)
    g3 = grid if grid.ndim == 3 else grid[np.newaxis]
    D, H, W = g3.shape
    nd = 2 if grid.ndim == 2 else 3
    out = np.empty((int(grid.max()) + 1, nd), dtype=np.float64)
    lib().orc_centers(_p(g3), C.c_int(D), C.c_int(H), C.c_int(W), C.c_int(grid.ndim == 2), _p(out))
    return out


# ------------------------------------------------------------------------------------------------
# graph cut
# ------------------------------------------------------------------------------------------------


def cut_general_graph(edges, edge_weights, unary_cost, pairwise_cost, n_iter=-1,
                      algorithm='expansion', return_energy=False):
    """CPU restatement of ``gco.cut_general_graph`` (gco-wrapper 3.0.x), expansion only"""
    assert algorithm == 'expansion'
    edges = np.ascontiguousarray(np.asarray(edges).reshape(-1, 2), dtype=np.int32)
    ew = np.ascontiguousarray(edge_weights, dtype=np.float64)
    un = np.ascontiguousarray(unary_cost, dtype=np.float64)
    pw = np.ascontiguousarray(pairwise_cost, dtype=np.float64)
    K, nc = un.shape
    if len(edges) and (edges[:, 0] >= edges[:, 1]).any():
        raise ValueError('edges must satisfy edges[:, 0] < edges[:, 1]')
    if pw.shape != (nc, nc) or (pw != pw.T).any():
        raise ValueError('Cost matrix not square or not symmetric')
    labels = np.empty(K, dtype=np.int32)
    e = lib().orc_cut_general_graph(_p(edges), C.c_int(len(edges)), _p(ew), _p(un), C.c_int(K),
                                    C.c_int(nc), _p(pw), C.c_int(n_iter), _p(labels))
    return (labels, e) if return_energy else labels


def gather_i32(lut, idx):
    lut = np.ascontiguousarray(lut, dtype=np.int32)
    idx32 = np.ascontiguousarray(idx, dtype=np.int32)
    out = np.empty(idx32.shape, dtype=np.int32)
    lib().orc_gather_i32(_p(lut), _p(idx32), C.c_size_t(idx32.size), _p(out))
    return out


def gather_f64(lut, idx):
    lut = np.ascontiguousarray(lut, dtype=np.float64)
    idx32 = np.ascontiguousarray(idx, dtype=np.int32)
    out = np.empty(idx32.shape + (lut.shape[1],), dtype=np.float64)
    lib().orc_gather_f64(_p(lut), C.c_int(lut.shape[1]), _p(idx32), C.c_size_t(idx32.size), _p(out))
    return out


def histogram_regions_labels_counts(slic, segm):
    """imsegm/labeling.py:208-247 on the oracle: float matrix [max(slic) + 1, max(segm) + 1] of pixel counts"""
    slic = np.ascontiguousarray(slic, dtype=np.int32)
    segm = np.ascontiguousarray(segm, dtype=np.int32)
    if slic.shape != segm.shape:
        raise TypeError('dimension does not agree')
    if segm.size and segm.min() < 0:
        raise ValueError('only positive labels are allowed')
    K, nb = int(slic.max()) + 1, int(segm.max()) + 1
    hist = np.zeros((K, nb), dtype=np.int64)
    lib().orc_label_hist(_p(slic), _p(segm), C.c_size_t(slic.size), C.c_int(K), C.c_int(nb), _p(hist))
    return hist.astype(np.float64)


def histogram_regions_labels_norm(slic, segm):
    """imsegm/labeling.py:250-280: rows divided by their sums (empty rows stay 0)"""
    hist = histogram_regions_labels_counts(slic, segm)
    sums = hist.sum(axis=1, keepdims=True)
    sums[sums == 0] = -1.
    hist = np.nan_to_num(hist / sums)
    hist[hist == 0] = 0
    return hist


def slic_gray3d_float32(vol, n_segments, compactness, sigma=1., spacing=(1., 1., 1.), max_iter=10, enforce_connectivity=True,
                        start_label=0, return_raw=False):
    """``skimage.segmentation.slic(vol_float32, ..., multichannel=False)`` as scikit-image 0.18 evaluates it: float32
    from end to end (see imsegm_oracle.c; HIP counterpart: the float32 section of csrc/volume.hip)."""
    L = lib()
    vol = np.ascontiguousarray(vol, dtype=np.float32)
    D, H, W = vol.shape
    spacing32 = np.asarray(spacing, dtype=np.float32)
    sig = np.array([sigma, sigma, sigma], dtype=np.float32) / spacing32           # float32 division, then float(sigma)
    taps = [gaussian_taps(float(s)) for s in sig]
    tap_args = []
    for t in taps:
        tap_args += [None, C.c_int(-1)] if t is None else [_p(t), C.c_int(len(t) - 1)]
    cent, steps = grid_centroids((D, H, W), n_segments)
    K = cent.shape[0]
    step = max(steps)
    isteps = [int(s if s is not None else 1) for _, s in regular_grid((D, H, W), K)]
    pre = np.empty((D, H, W), dtype=np.float32)
    nearest = np.empty((D, H, W), dtype=np.int32)
    L.orc_slic_gray3d_f32(_p(vol), C.c_int(D), C.c_int(H), C.c_int(W), *tap_args, C.c_double(1.0 / compactness), C.c_int(K),
                          _p(np.ascontiguousarray(cent[:, :3], dtype=np.float64)), C.c_int(isteps[0]), C.c_int(isteps[1]),
                          C.c_int(isteps[2]), C.c_float(step), _p(np.asarray(spacing32, dtype=np.float64)), C.c_int(max_iter),
                          _p(pre), _p(nearest))
    labels = nearest + start_label
    if return_raw or not enforce_connectivity:
        return labels.astype(np.int64)
    segment_size = D * H * W / K
    out = np.empty_like(labels)
    L.orc_enforce_connectivity(_p(labels), C.c_int(D), C.c_int(H), C.c_int(W), C.c_long(int(0.5 * segment_size)),
                               C.c_long(int(3 * segment_size)), C.c_int(start_label), _p(out))
    return out.astype(np.int64)


def cut_grid_graph(unary_cost, pairwise_cost, cost_v, cost_h, n_iter=-1, algorithm='expansion'):
    """CPU restatement of ``gco.cut_grid_graph`` (gco-wrapper 3.0.x, pygco.py: the vertical edges followed by the horizontal
    ones of a general graph, one down-weight factor over all of them), called at imsegm/region_growing.py:248"""
    unary_cost = np.asarray(unary_cost, dtype=np.float64)
    height, width, n_labels = unary_cost.shape
    index = np.arange(height * width, dtype=np.int32).reshape(height, width)
    edges = np.concatenate([np.stack([index[:-1].ravel(), index[1:].ravel()], axis=1),
                            np.stack([index[:, :-1].ravel(), index[:, 1:].ravel()], axis=1)], axis=0)
    weights = np.concatenate([np.asarray(cost_v, dtype=np.float64).ravel(), np.asarray(cost_h, dtype=np.float64).ravel()])
    return cut_general_graph(edges, weights, unary_cost.reshape(height * width, n_labels), pairwise_cost, n_iter=n_iter,
                             algorithm=algorithm)
