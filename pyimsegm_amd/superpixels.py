"""Superpixel stage: SLIC clustering, region adjacency graph, superpixel centres.

Host-side mirror of the reference module ``imsegm/superpixels.py`` (same function names, argument
meaning and return types).  Where the reference calls ``skimage.segmentation.slic`` /
``skimage.measure.regionprops`` or loops over pixels in Python, this module calls the HIP kernels
of ``libimsegm_hip.so``.
"""
import logging

import numpy as np

from pyimsegm_amd import _hip

#: spacing among neighbouring pixels in axes X, Y, Z
IMAGE_SPACING = (1, 1, 1)
#: first label produced by SLIC -- scikit-image 0.18 semantics (``start_label=None`` -> 0), the
#: newest release line that still accepts the reference's ``multichannel=`` keyword
SLIC_START_LABEL = 0
#: number of k-means sweeps (``skimage.segmentation.slic(max_iter=10)`` default)
SLIC_MAX_ITER = 10


def _as_rgb(img):
    img = np.asarray(img)
    if img.ndim == 2:  # replicate the gray channel, reference superpixels.py:50-51
        img = np.repeat(img[:, :, np.newaxis], 3, axis=2)
    if img.ndim != 3 or img.shape[2] != 3:
        raise ValueError('expected a gray or RGB 2D image, got shape %r' % (img.shape, ))
    return img


def _slic_params(shape2d, sp_size, relative_compact):
    """parameter mapping of the reference, superpixels.py:57-58"""
    nb_pixels = np.prod(shape2d)
    return int(nb_pixels / (sp_size**2)), (sp_size * relative_compact)**1.5


def _open_session(img, reuse=False):
    """upload an image for the device-resident pipeline; returns (session, normalize_mode)

    ``reuse``: take the idle session of the same size that this thread's context keeps from an earlier image
    (its device buffers are recycled, only the pixels are uploaded); give it back with :func:`_release_session`."""
    img = _as_rgb(img)
    mode = 2  # min-max scale unless min == 0 and max == 1 (superpixels.py:53-54), decided on device
    if img.dtype not in (np.uint8, np.float32, np.float64):
        # numpy evaluates ``img - img.min()`` in the image's own dtype: do that step here so that
        # exotic integer types wrap exactly as in the reference
        if img.min() != 0. or img.max() != 1.:
            img = (img - img.min()) / float(img.max() - img.min())
        else:
            img = img.astype(np.float64)
        mode = 0
    elif img.dtype == np.float32:
        # the reference divides float32 by a Python float -> float32 result, then skimage widens it
        if img.min() != 0. or img.max() != 1.:
            img = ((img - img.min()) / float(img.max() - img.min())).astype(np.float64)
        else:
            img = img.astype(np.float64)
        mode = 0
    sess = None
    if reuse:
        idle = _hip.default_context().idle_sessions
        sess = idle.pop(img.shape[:2], None)
    if sess is None:
        sess = _hip.Image2D(img.shape[0], img.shape[1])
    sess.upload(img)
    return sess, mode


def _release_session(sess):
    """keep a session for the next image of the same size on this thread (one per size), else close it"""
    idle = sess.ctx.idle_sessions
    if sess.shape in idle or sess.ctx is not _hip.default_context():
        sess.close()
        return
    if len(sess.shape) == 3:
        # a volume session owns ~70 bytes of device memory per voxel (75 GB at 64 x 4096 x 4096): at most ONE idle volume
        # session per context, whatever its shape
        _evict_idle_volumes(idle)
    idle[sess.shape] = sess


def _evict_idle_volumes(idle, keep=None):
    """close the idle volume sessions of a context (all shapes but ``keep``)"""
    for shape in [s for s in idle if len(s) == 3 and s != keep]:
        idle.pop(shape).close()


def _run_slic(sess, mode, sp_size, relative_compact, slico=False):
    n_seg, compact = _slic_params(sess.shape, sp_size, relative_compact)
    logging.debug('Starting SLIC with params NB=%i & compat=%f for image %r', n_seg, compact, sess.shape)
    if n_seg < 1:
        raise ValueError('superpixel size %r is larger than the image %r' % (sp_size, sess.shape))
    return sess.slic(n_seg, compact, sigma=1., normalize=mode, max_iter=SLIC_MAX_ITER, enforce_connectivity=True,
                     start_label=SLIC_START_LABEL, slic_zero=slico)


def segment_slic_img2d(img, sp_size=50, relative_compact=0.1, slico=False):
    """ SLIC superpixels of a 2D gray / colour image

    :param ndarray img: input image, H x W or H x W x 3
    :param int sp_size: initial superpixel size (edge length in pixels)
    :param float relative_compact: regularisation in (0, 1); 0 free-form, 1 nearly square
    :param bool slico: parameter-free SLICO variant (``slic_zero=True`` of skimage; exact fp64 sweeps on the device)
    :return ndarray: int64 label map H x W
    """
    logging.debug('Init SLIC superpixels 2d RGB clustering with params size=%i and regul=%f for image dims %r',
                  sp_size, relative_compact, np.shape(img))
    sess, mode = _open_session(img)
    _run_slic(sess, mode, sp_size, relative_compact, slico=slico)
    logging.debug('SLIC finished')
    labels = sess.get_labels()
    sess.close()
    return labels


def _slic3d_params(shape3d, sp_size, relative_compact, space):
    """parameter mapping of the reference, superpixels.py:92-97 (float32 spacing, truncated compactness)"""
    nb_pixels = np.prod(shape3d)
    sp_vol = np.prod(sp_size / np.asarray(space, dtype=np.float32) * min(space))
    return int(nb_pixels / sp_vol), int((sp_vol * relative_compact)**1.5)


def _open_volume(im, reuse=False):
    """``reuse``: take the idle session of the same shape that this thread's context keeps from an earlier volume (a volume
    session owns ~70 bytes of device memory per voxel; allocating and freeing them costs seconds at 64 x 4096 x 4096); give it back
    with :func:`_release_session`"""
    im = np.asarray(im)
    if im.ndim != 3:
        raise ValueError('expected a 3D gray volume, got shape %r' % (im.shape, ))
    idle = _hip.default_context().idle_sessions
    sess = idle.pop(tuple(im.shape), None) if reuse else None
    if sess is None:
        _evict_idle_volumes(idle)          # the memory of an idle volume of another shape is needed now
        sess = _hip.Volume3D(*im.shape)
    try:
        return sess.upload(im)
    except _hip.HipUnavailableError:
        sess.close()
        raise
    except Exception:
        # (device allocations happen on first use: upload, slic) out of memory with idle sessions around: give every idle
        # session of this context back and try once more
        sess.close()
        if not idle:
            raise
        for shape in list(idle):
            idle.pop(shape).close()
        sess = _hip.Volume3D(*im.shape)
        try:
            return sess.upload(im)
        except Exception:
            sess.close()
            raise


def _run_slic3d(sess, sp_size, relative_compact, space):
    n_seg, compact = _slic3d_params(sess.shape, sp_size, relative_compact, space)
    logging.debug('Starting SLIC superpixels clustering with params NB=%i and compat=%f and spacing=%r', n_seg, compact,
                  space)
    if n_seg < 1:
        raise ValueError('superpixel size %r is larger than the volume %r' % (sp_size, sess.shape))
    if compact < 1:
        raise ZeroDivisionError('SLIC compactness truncates to 0 (sp_size=%r, relative_compact=%r)'
                                % (sp_size, relative_compact))
    sess.slic(n_seg, compact, sigma=1., spacing=space, max_iter=SLIC_MAX_ITER, enforce_connectivity=True,
              start_label=SLIC_START_LABEL)
    # fix of unconnected segments, superpixels.py:111 (skimage.measure.label)
    return sess.label_cc()


def segment_slic_img3d_gray(im, sp_size=50, relative_compact=0.1, space=IMAGE_SPACING):
    """ SLIC supervoxels of a 3D gray volume followed by connected-component relabelling

    :param ndarray im: input 3D gray-scale image
    :param int sp_size: initial supervoxel size
    :param float relative_compact: regularisation in (0, 1)
    :param tuple(int,int,int) space: voxel spacing per axis
    :return ndarray: int64 label map of the volume's shape (labels from 1, as ``measure.label`` gives)
    """
    logging.debug('Init SLIC superpixels 3d Gray clustering with params size=%i and regul=%f for image dims %r',
                  sp_size, relative_compact, np.shape(im))
    sess = _open_volume(im)
    _run_slic3d(sess, sp_size, relative_compact, space)
    labels = sess.get_labels()
    sess.close()
    return labels


def make_graph_segment_connect_edges(vertices, all_edges):
    """ unique undirected edges among vertex pairs (numpy helper kept for API compatibility)

    :param ndarray vertices: sorted unique labels
    :param ndarray all_edges: N x 2 pairs of indexes into ``vertices``
    :return tuple(ndarray,list): vertices, edges ``[[a, b], ...]`` ordered by (b, a)
    """
    all_edges = np.asarray(all_edges)
    pairs = np.sort(all_edges[all_edges[:, 0] != all_edges[:, 1], :], axis=1)
    nb = len(vertices)
    codes = np.unique(pairs[:, 0] + nb * pairs[:, 1])
    edges = [[vertices[int(c % nb)], vertices[int(c // nb)]] for c in codes]
    return vertices, edges


def get_segment_diffs_2d_conn4(grid):
    """ all right / down neighbour label pairs of a 2D label map """
    grid = np.asarray(grid)
    right = np.stack([grid[:, :-1].ravel(), grid[:, 1:].ravel()], axis=1)
    down = np.stack([grid[:-1, :].ravel(), grid[1:, :].ravel()], axis=1)
    return np.vstack([right, down])


def get_segment_diffs_3d_conn6(grid):
    """ all 6-connected neighbour label pairs of a 3D label map """
    grid = np.asarray(grid)
    below = np.stack([grid[:-1, :, :].ravel(), grid[1:, :, :].ravel()], axis=1)
    down = np.stack([grid[:, :-1, :].ravel(), grid[:, 1:, :].ravel()], axis=1)
    right = np.stack([grid[:, :, :-1].ravel(), grid[:, :, 1:].ravel()], axis=1)
    return np.vstack([below, right, down])


def _session_for_labels(grid):
    grid = np.asarray(grid)
    if grid.ndim not in (2, 3):
        raise ValueError('2D or 3D label map expected')
    if grid.size and grid.min() < 0:
        raise ValueError('labels must be non-negative')
    if grid.ndim == 3:
        return _hip.Volume3D(*grid.shape).set_labels(grid)
    return _hip.Image2D(grid.shape[0], grid.shape[1]).set_labels(grid)


def _graph_from_session(sess):
    edges, centres, present = sess.graph()
    vertices = np.flatnonzero(present)
    return vertices, edges, centres, present


def make_graph_segm_connect_grid2d_conn4(grid):
    """ region adjacency graph (4-connectivity) of a 2D label map

    :param ndarray grid: segmentation
    :return tuple(ndarray,list): unique labels, list of edges ``[a, b]`` with a < b ordered by (b, a)
    """
    logging.debug('make graph segment connect edges - 2d conn4')
    sess = _session_for_labels(grid)
    vertices, edges, _, _ = _graph_from_session(sess)
    sess.close()
    return vertices, edges.tolist()


def make_graph_segm_connect_grid3d_conn6(grid):
    """ region adjacency graph (6-connectivity) of a 3D label map

    :param ndarray grid: segmentation
    :return tuple(ndarray,list): unique labels, list of edges ``[a, b]`` with a < b ordered by (b, a)
    """
    logging.debug('make graph segment connect edges - 3d conn6')
    grid = np.asarray(grid)
    if grid.ndim != 3:
        raise ValueError('3D label map expected')
    sess = _session_for_labels(grid)
    vertices, edges, _, _ = _graph_from_session(sess)
    sess.close()
    return vertices, edges.tolist()


def superpixel_centers(segments):
    """ centre of mass (row, col[, ...]) of every superpixel; ``[-1] * ndim`` for unused labels

    :param ndarray segments: label map
    :return list: per label a tuple (2D) / list (3D) of coordinates
    """
    segments = np.asarray(segments)
    logging.debug('compute centers for %d superpixels', segments.max())
    if segments.ndim == 2:
        sess = _session_for_labels(segments)
        _, _, centres, present = _graph_from_session(sess)
        sess.close()
        return [tuple(c.tolist()) if ok else [-1] * 2 for c, ok in zip(centres, present)]
    if segments.ndim == 3:
        sess = _session_for_labels(segments)
        _, _, centres, present = _graph_from_session(sess)
        sess.close()
        return [c.tolist() if ok else [-1] * 3 for c, ok in zip(centres, present)]
    logging.error('not supported image dim: %r', segments.shape)
    return [[-1] * segments.ndim for _ in range(int(segments.max()) + 1)]


def get_neighboring_segments(edges):
    """ neighbour lists from an edge list
    """
    neighbours = [[] for _ in range(int(np.max(edges)) + 1)]
    for a, b in edges:
        neighbours[a].append(b)
        neighbours[b].append(a)
    return neighbours
