"""Superpixel x annotation histograms -- the part of the reference module ``imsegm/labeling.py`` that sits on
the supervised SLIC -> features -> classifier -> GraphCut path (``histogram_regions_labels_counts`` :208,
``histogram_regions_labels_norm`` :250, called at ``imsegm/pipelines.py:284``).

The reference counts with a per-pixel Python loop (``labeling.py:244-245``); here the pairs are counted by the
HIP kernel ``k_label_hist`` (``csrc/stats.hip``) through ``imsegm_image2d_label_hist``.  There is no CPU
fallback: without the HIP library the calls raise.
"""
import numpy as np

from pyimsegm_amd import _hip
from pyimsegm_amd.utilities import ImageDimensionError


def _flat2d(arr):
    """any-dimensional label array as rows x columns (the histogram only sees the flat order)"""
    arr = np.asarray(arr)
    if arr.ndim == 0:
        return arr.reshape(1, 1)
    return arr.reshape(-1, arr.shape[-1]) if arr.ndim != 2 else arr


def histogram_regions_labels_counts(slic, segm, _session=None):
    """ histogram of overlapping regions between two segmentations,
    the typical usage is labelling superpixels from an annotation

    :param ndarray slic: superpixel map
    :param ndarray segm: annotation of the same shape, non-negative labels
    :param _session: (internal) device session that already holds ``slic`` as its label map
    :return ndarray: float matrix, rows = superpixels ``0..max(slic)``, columns = labels ``0..max(segm)``
    """
    segm = np.asarray(segm)
    if _session is None:
        slic = np.asarray(slic)
        if slic.shape != segm.shape:
            raise ImageDimensionError('dimension does not agree')
    elif tuple(_session.shape) != segm.shape:
        raise ImageDimensionError('dimension does not agree')
    if segm.size and segm.min() < 0:
        raise ValueError('only positive labels are allowed')
    nb_annot = int(segm.max()) + 1 if segm.size else 1
    if _session is not None:
        counts = _session.label_hist(segm, nb_annot)
    else:
        slic2d = _flat2d(slic)
        if slic2d.size and slic2d.min() < 0:
            raise ValueError('only positive superpixel labels are allowed')
        sess = _hip.Image2D(slic2d.shape[0], slic2d.shape[1]).set_labels(slic2d)
        try:
            counts = sess.label_hist(_flat2d(segm), nb_annot)
        finally:
            sess.close()
    return counts.astype(np.float64)


def histogram_regions_labels_norm(slic, segm, _session=None):
    """ normalised histogram of overlapping regions between two segmentations: the relative overlap of every
    superpixel with every annotation label (rows of superpixels without pixels stay zero)

    :param ndarray slic: superpixel map
    :param ndarray segm: annotation of the same shape
    :return ndarray: rows = superpixels, columns = labels; every non-empty row sums to one
    """
    shape = tuple(_session.shape) if _session is not None else np.shape(slic)
    if shape != np.shape(segm):
        raise ImageDimensionError('dimension of SLIC %r and segm %r should match' % (shape, np.shape(segm)))
    matrix_hist = histogram_regions_labels_counts(slic, segm, _session=_session)
    region_sums = matrix_hist.sum(axis=1, keepdims=True)
    region_sums[region_sums == 0] = -1.            # no division by zero
    matrix_hist = np.nan_to_num(matrix_hist / region_sums)
    matrix_hist[matrix_hist == 0] = 0              # no negative zeros
    return matrix_hist


def assume_bg_on_boundary(segm, bg_label=0, boundary_size=1):
    """ swap labels such that the background label is the one that dominates the image boundary
    (reference ``labeling.py:719-753``, called by the driver at ``run_segm_slic_model_graphcut.py:373,422``)

    2-D integer label images whose values fit int32 go through the device (``imsegm_assume_bg_on_boundary``: the histogram
    of the four border strips and the label exchange are two kernels on the uploaded map); anything else -- other
    dimensions, float labels -- through the numpy statements of the reference.

    :param ndarray segm: label image
    :param int bg_label: the label the background is to carry
    :param float boundary_size: width of the border that is looked at
    :return ndarray: segmentation with the boundary label and ``bg_label`` exchanged
    """
    arr = np.asarray(segm)
    size = int(boundary_size)
    if arr.ndim == 2 and arr.size and 0 < size <= min(arr.shape) and arr.dtype.kind in 'iu' and 0 <= int(bg_label) < 2**31 \
            and int(arr.min()) >= 0 and int(arr.max()) < 2**28:
        # (negative labels wrap around in the reference's ``np.array(lut)[segm]``, values of 2^28 and more do not fit the
        # device's border histogram: both take the numpy statements below)
        height, width = arr.shape
        # the four strips of data_io.py:1026 (0 < size <= min(shape): the only sizes np.hstack accepts there)
        rows_top, cols_left = slice(None, size).indices(height), slice(None, size).indices(width)
        rows_bottom, cols_right = slice(-size, None).indices(height), slice(-size, None).indices(width)
        strips = np.array([rows_top[0], max(rows_top[1], rows_top[0]), 0, width,
                           0, height, cols_left[0], max(cols_left[1], cols_left[0]),
                           rows_bottom[0], max(rows_bottom[1], rows_bottom[0]), 0, width,
                           0, height, cols_right[0], max(cols_right[1], cols_right[0])], dtype=np.int32)
        work = np.ascontiguousarray(arr, dtype=np.int32)
        if work is arr or np.shares_memory(work, arr):
            work = work.copy()
        _hip.assume_bg_on_boundary(work, strips, bg_label)
        # the reference indexes a Python list of ints: the result is an int64 array
        return work.astype(np.int64)
    from pyimsegm_amd.utilities.data_io import get_image2d_boundary_color
    on_border = int(get_image2d_boundary_color(segm, size=boundary_size))
    present = np.unique(segm)
    if on_border not in present:                   # (cannot happen for integer labels; the reference's statements in that case)
        segm[segm == on_border] = bg_label
        return segm
    # the exchange as a look-up table over 0 .. max(label, bg_label) (when the background label is not in use the reference's
    # table may be too short for it)
    lut = np.arange(max(int(present.max()), int(bg_label)) + 1)
    lut[[on_border, bg_label]] = bg_label, on_border
    return lut[segm]
