"""Descriptor stage: per-superpixel colour / texture statistics.

Host-side mirror of the hot-path part of the reference module ``imsegm/descriptors.py`` (lines
1-1285: colour statistics, Leung-Malik texture bank, feature-set selection).  The segmented
reductions the reference delegates to ``imsegm/features_cython.pyx`` run as HIP kernels
(``csrc/stats.hip``); the function names ``cython_*`` are kept so that callers of the reference do
not change, ``hip_*`` are the same functions under an honest name.

Host (numpy / scipy) code remains where the reference itself is numpy / scipy: the ``numpy_*``
alternatives, the median / mean-gradient statistics and the Leung-Malik filter responses
(``scipy.ndimage`` convolutions, exactly as ``descriptors.py:903-1106``) -- those are not yet on the
HIP path ("what runs where").
"""
import functools
import itertools
import logging

import numpy as np
from scipy import ndimage
from sklearn import preprocessing

from pyimsegm_amd import _hip
from pyimsegm_amd.utilities import ImageDimensionError

#: kept for API compatibility: the driver script assigns it (run_segm_slic_model_graphcut.py:59).
#: It no longer selects an implementation: the native path is always the HIP library.
USE_CYTHON = True
#: the segmented reductions run on the GPU through libimsegm_hip.so (there is no CPU fallback)
USE_HIP = True

#: define all available statistic computed on superpixels
NAMES_FEATURE_FLAGS = ('mean', 'std', 'energy', 'median', 'meanGrad')
#: define sigmas for Leung-Malik filter bank
DEFAULT_FILTERS_SIGMAS = (np.sqrt(2), 2, 2 * np.sqrt(2), 4)
#: define small list/range of sigmas for Leung-Malik filter bank
SHORT_FILTERS_SIGMAS = (np.sqrt(2), 2, 4)
#: define the richest version of computed superpixel features
FEATURES_SET_ALL = {
    'color': ('mean', 'std', 'energy', 'median', 'meanGrad'),
    'tLM': ('mean', 'std', 'energy', 'median', 'meanGrad'),
}
#: define basic color features for superpixels
FEATURES_SET_COLOR = {'color': ('mean', 'std', 'energy')}
#: define basic texture features (complete LM filter bank) for superpixels
FEATURES_SET_TEXTURE = {'tLM': ('mean', 'std', 'energy')}
#: define basic texture features (small LM filter bank) for superpixels
FEATURES_SET_TEXTURE_SHORT = {'tLM_short': ('mean', 'std', 'energy')}
#: define circular diameters for computing label histogram
HIST_CIRCLE_DIAGONALS = (10, 20, 30, 40, 50)
#: maximal response is bounded by fix number to prevent overflowing (for LM filter bank)
MAX_SIGNAL_RESPONSE = 1.e6


# ------------------------------------------------------------------------------------------------
# input validation (messages as the reference, descriptors.py:128-206)
# ------------------------------------------------------------------------------------------------


def _check_color_image_segm(image, segm):
    """ image - segmentation compatibility for colour images

    >>> _check_color_image_segm(np.zeros((125, 150, 3)), np.zeros((150, 125)))  # doctest: +ELLIPSIS
    Traceback (most recent call last):
    ...
    pyimsegm_amd.utilities.ImageDimensionError: ndarrays - image and segmentation do not match (125, 150, 3) vs (150, 125)
    """
    if image.shape[:2] != segm.shape:
        raise ImageDimensionError('ndarrays - image and segmentation do not match %r vs %r' % (image.shape, segm.shape))
    return True


def _check_gray_image_segm(image, segm):
    """ image - segmentation compatibility for gray images / volumes

    >>> _check_gray_image_segm(np.zeros((125, 150)), np.zeros((150, 125)))  # doctest: +ELLIPSIS
    Traceback (most recent call last):
    ...
    pyimsegm_amd.utilities.ImageDimensionError: ndarrays - image and segmentation do not match (125, 150) vs (150, 125)
    """
    if image.shape != segm.shape:
        raise ImageDimensionError('ndarrays - image and segmentation do not match %r vs %r' % (image.shape, segm.shape))
    return True


def _check_color_image(image):
    """ the image has to be H x W x 3

    >>> _check_color_image(np.zeros((200, 250, 1)))  # doctest: +ELLIPSIS
    Traceback (most recent call last):
    ...
    pyimsegm_amd.utilities.ImageDimensionError: image is not RGB with dims (200, 250, 1)
    """
    if image.ndim != 3 or image.shape[2] != 3:
        raise ImageDimensionError('image is not RGB with dims %s' % repr(image.shape))
    return True


def _check_unrecognised_feature_group(feature_flags):
    """ report (not raise) unknown feature groups

    >>> _check_unrecognised_feature_group({'color': [], 'texture': []})
    ['texture']
    """
    unknown = [k for k in feature_flags if not (k.startswith('color') or k.startswith('tLM'))]
    if unknown:
        logging.warning('unrecognised following feature groups: %r', unknown)
    return unknown


def _check_unrecognised_feature_names(feature_flags):
    """ report (not raise) unknown statistic names

    >>> _check_unrecognised_feature_names(['mean', 'average'])
    ['average']
    """
    unknown = [k for k in feature_flags if k not in NAMES_FEATURE_FLAGS]
    if unknown:
        logging.warning('unrecognised following feature names: %r', unknown)
    return unknown

def _finished(features, names):
    """the closing step every feature table of the reference goes through: negative zeros become zeros
    (``features[features == 0] = 0``, descriptors.py:860,1034,1102,1165,1265) and the column count is checked against the names"""
    features[features == 0] = 0
    if features.shape[1] != len(names):
        raise ValueError('features: %r and names %r' % (features.shape, names))
    return features, names



# ------------------------------------------------------------------------------------------------
# HIP segmented statistics, colour 2D  (reference: cython_img2d_color_*, descriptors.py:209-296)
# ------------------------------------------------------------------------------------------------


def _stats_session(img, seg):
    img = np.asarray(img)
    seg = np.asarray(seg)
    _check_color_image_segm(img, seg)
    _check_color_image(img)
    return _hip.Image2D(seg.shape[0], seg.shape[1]).upload(img).set_labels(seg)


def hip_img2d_color_mean(img, seg):
    """ mean colour per superpixel (float32 staging, 64-bit accumulation as features_cython.pyx:81-98)

    :param ndarray img: input RGB image H x W x 3
    :param ndarray seg: segmentation H x W
    :return ndarray: np.array<nb_lbs, 3>

    >>> image = np.zeros((2, 10, 3))
    >>> image[:, 2:6, 0] = 1
    >>> image[:, 3:7, 1] = 3
    >>> image[:, 4:9, 2] = 2
    >>> segm = np.array([[0, 0, 0, 0, 0, 1, 1, 1, 1, 1],
    ...                  [0, 0, 0, 0, 0, 1, 1, 1, 1, 1]])
    >>> hip_img2d_color_mean(image, segm)  # doctest: +SKIP
    array([[0.6, 1.2, 0.4],
           [0.2, 1.2, 1.6]])
    """
    logging.debug('HIP: computing Colour means for image %r & segm %r', np.shape(img), np.shape(seg))
    sess = _stats_session(img, seg)
    mean, _, _ = sess.color_stats(mean=True, energy=False, var=False)
    sess.close()
    return mean


def hip_img2d_color_energy(img, seg):
    """ mean squared colour per superpixel (features_cython.pyx:101-119) """
    logging.debug('HIP: computing Colour energy for image %r & segm %r', np.shape(img), np.shape(seg))
    sess = _stats_session(img, seg)
    _, energy, _ = sess.color_stats(mean=False, energy=True, var=False)
    sess.close()
    return energy


def hip_img2d_color_std(img, seg, means=None):
    """ colour standard deviation per superpixel (features_cython.pyx:122-141, descriptors.py:264-296)

    The variance kernel subtracts the float32-rounded means, exactly as the reference does.
    ``means`` is accepted for API compatibility; the means are (re)computed on the device, which is
    what the reference does when ``means`` is None and what its callers pass anyway.
    """
    logging.debug('HIP: computing Colour STD for image %r & segm %r', np.shape(img), np.shape(seg))
    sess = _stats_session(img, seg)
    _, _, var = sess.color_stats(mean=False, energy=False, var=True)
    sess.close()
    return np.sqrt(var)


cython_img2d_color_mean = hip_img2d_color_mean
cython_img2d_color_energy = hip_img2d_color_energy
cython_img2d_color_std = hip_img2d_color_std

# ------------------------------------------------------------------------------------------------
# numpy alternatives (reference: numpy_img2d_color_*, descriptors.py:299-455) -- vectorised, float64
# ------------------------------------------------------------------------------------------------


def _segmented_sums(values, seg, nb_labels):
    """sum of ``values`` (N x C float64) per label -> (nb_labels x C, counts)"""
    flat = seg.ravel()
    counts = np.bincount(flat, minlength=nb_labels).astype(np.float64)
    sums = np.stack([np.bincount(flat, weights=values[:, c], minlength=nb_labels) for c in range(values.shape[1])],
                    axis=1)
    return sums, counts


def _safe_div(sums, counts):
    counts = counts.copy()
    counts[counts == 0] = -1
    return sums / counts[:, np.newaxis]


def numpy_img2d_color_mean(img, seg):
    """ colour means by numpy (float64 throughout)

    >>> image = np.zeros((2, 10, 3))
    >>> image[:, 2:6, 0] = 1
    >>> image[:, 3:8, 1] = 3
    >>> image[:, 4:9, 2] = 2
    >>> segm = np.array([[0, 0, 0, 0, 0, 1, 1, 1, 1, 1],
    ...                  [0, 0, 0, 0, 0, 1, 1, 1, 1, 1]])
    >>> numpy_img2d_color_mean(image, segm).tolist()
    [[0.6, 1.2, 0.4], [0.2, 1.8, 1.6]]
    """
    img, seg = np.asarray(img), np.asarray(seg)
    _check_color_image_segm(img, seg)
    nb = int(np.max(seg)) + 1
    sums, counts = _segmented_sums(img.reshape(-1, 3).astype(np.float64), seg, nb)
    return _safe_div(sums, counts)


def numpy_img2d_color_std(img, seg, means=None):
    """ colour STD by numpy

    >>> image = np.zeros((2, 10, 3))
    >>> image[:, 2:6, 0] = 1
    >>> image[:, 3:8, 1] = 3
    >>> image[:, 4:9, 2] = 2
    >>> segm = np.array([[0, 0, 0, 0, 0, 1, 1, 1, 1, 1],
    ...                  [0, 0, 0, 0, 0, 1, 1, 1, 1, 1]])
    >>> np.round(numpy_img2d_color_std(image, segm), 8).tolist()
    [[0.48989795, 1.46969385, 0.8], [0.4, 1.46969385, 0.8]]
    """
    img, seg = np.asarray(img), np.asarray(seg)
    _check_color_image_segm(img, seg)
    if means is None:
        means = numpy_img2d_color_mean(img, seg)
    nb = int(np.max(seg)) + 1
    if len(means) < nb:
        raise ValueError('number of means (%i) should be equal to number of labels (%i)' % (len(means), nb))
    diff = img.reshape(-1, 3).astype(np.float64) - np.asarray(means)[seg.ravel()]
    sums, counts = _segmented_sums(diff**2, seg, nb)
    variations = _safe_div(sums, counts)
    variations[variations == 0] = 0
    return np.sqrt(variations)


def numpy_img2d_color_energy(img, seg):
    """ colour energy by numpy (squares evaluated in float64: the reference's uint8 wrap-around of
    ``img[i, j, :]**2``, descriptors.py:410, is NOT reproduced)

    >>> image = np.zeros((2, 10, 3))
    >>> image[:, 2:6, 0] = 1
    >>> image[:, 3:8, 1] = 3
    >>> image[:, 4:9, 2] = 2
    >>> segm = np.array([[0, 0, 0, 0, 0, 1, 1, 1, 1, 1],
    ...                  [0, 0, 0, 0, 0, 1, 1, 1, 1, 1]])
    >>> numpy_img2d_color_energy(image, segm).tolist()
    [[0.6, 3.6, 0.8], [0.2, 5.4, 3.2]]
    """
    img, seg = np.asarray(img), np.asarray(seg)
    _check_color_image_segm(img, seg)
    nb = int(np.max(seg)) + 1
    sums, counts = _segmented_sums(img.reshape(-1, 3).astype(np.float64)**2, seg, nb)
    return _safe_div(sums, counts)


def _segmented_median(values, flat_seg, nb_labels):
    """median per label of a 1D value array; NaN for labels without samples (np.median([]))"""
    order = np.lexsort((values, flat_seg))
    sv, sl = values[order], flat_seg[order]
    counts = np.bincount(flat_seg, minlength=nb_labels)
    starts = np.concatenate([[0], np.cumsum(counts)[:-1]])
    out = np.full(nb_labels, np.nan)
    ok = counts > 0
    lo = starts[ok] + (counts[ok] - 1) // 2
    hi = starts[ok] + counts[ok] // 2
    out[ok] = 0.5 * (sv[lo] + sv[hi])
    del sl
    return out


def numpy_img2d_color_median(img, seg):
    """ colour median by numpy

    >>> image = np.zeros((2, 10, 3))
    >>> image[:, 2:6, 0] = 1
    >>> image[:, 3:8, 1] = 3
    >>> image[:, 4:9, 2] = 2
    >>> segm = np.array([[0, 0, 0, 0, 1, 1, 1, 1, 1, 1],
    ...                  [0, 0, 0, 0, 1, 1, 1, 1, 1, 1]])
    >>> numpy_img2d_color_median(image, segm).tolist()
    [[0.5, 0.0, 0.0], [0.0, 3.0, 2.0]]
    """
    img, seg = np.asarray(img), np.asarray(seg)
    _check_color_image_segm(img, seg)
    nb = int(np.max(seg)) + 1
    flat = seg.ravel()
    vals = img.reshape(-1, 3).astype(np.float64)
    return np.stack([_segmented_median(vals[:, c], flat, nb) for c in range(3)], axis=1)


# ------------------------------------------------------------------------------------------------
# gray 3D statistics (reference: cython_img3d_gray_* / numpy_img3d_gray_*, descriptors.py:458-702)
# ------------------------------------------------------------------------------------------------


def _gray_sums(values, seg):
    nb = int(np.max(seg)) + 1
    flat = np.asarray(seg).ravel()
    counts = np.bincount(flat, minlength=nb)
    sums = np.bincount(flat, weights=np.asarray(values, dtype=np.float64).ravel(), minlength=nb)
    out = np.zeros(nb)
    ok = counts > 0
    out[ok] = sums[ok] / counts[ok]
    return out


def _gray_stats_session(img, seg):
    img, seg = np.asarray(img), np.asarray(seg)
    _check_gray_image_segm(img, seg)
    return _hip.Volume3D(*seg.shape).upload(img).set_labels(seg)


def hip_img3d_gray_mean(img, seg):
    """ mean intensity per supervoxel (float32 staging, 64-bit accumulation as features_cython.pyx:144-166)

    :param ndarray img: gray volume D x H x W
    :param ndarray seg: segmentation D x H x W
    :return ndarray: np.array<nb_lbs>

    >>> image = np.zeros((2, 3, 8))
    >>> image[0, :, 2:6] = 1
    >>> image[1, :, 3:7] = 3
    >>> segm = np.array([[[0, 0, 0, 0, 1, 1, 1, 1]] * 3,
    ...                  [[2, 2, 2, 2, 3, 3, 3, 3]] * 3])
    >>> hip_img3d_gray_mean(image, segm).tolist()  # doctest: +SKIP
    [0.5, 0.5, 0.75, 2.25]
    """
    logging.debug('HIP: computing Gray means for image %r and segm %r', np.shape(img), np.shape(seg))
    sess = _gray_stats_session(img, seg)
    mean, _, _ = sess.gray_stats(mean=True, energy=False, var=False)
    sess.close()
    return mean


def hip_img3d_gray_energy(img, seg):
    """ mean squared intensity per supervoxel (features_cython.pyx:169-191)

    >>> image = np.zeros((2, 3, 8))
    >>> image[0, :, 2:6] = 1
    >>> image[1, :, 3:7] = 3
    >>> segm = np.array([[[0, 0, 0, 0, 1, 1, 1, 1]] * 3,
    ...                  [[2, 2, 2, 2, 3, 3, 3, 3]] * 3])
    >>> hip_img3d_gray_energy(image, segm).tolist()  # doctest: +SKIP
    [0.5, 0.5, 2.25, 6.75]
    """
    logging.debug('HIP: computing Gray energy for image %r and segm %r', np.shape(img), np.shape(seg))
    sess = _gray_stats_session(img, seg)
    _, energy, _ = sess.gray_stats(mean=False, energy=True, var=False)
    sess.close()
    return energy


def hip_img3d_gray_std(img, seg, mean=None):
    """ intensity STD per supervoxel (features_cython.pyx:194-219; float32-rounded mean, descriptors.py:547)

    ``mean`` is accepted for API compatibility; the means are (re)computed on the device, which is
    what the reference does when ``mean`` is None and what its callers pass anyway.

    >>> image = np.zeros((2, 3, 8))
    >>> image[0, :, 2:6] = 1
    >>> image[1, :, 3:7] = 3
    >>> segm = np.array([[[0, 0, 0, 0, 1, 1, 1, 1]] * 3,
    ...                  [[2, 2, 2, 2, 3, 3, 3, 3]] * 3])
    >>> np.round(hip_img3d_gray_std(image, segm), 4).tolist()  # doctest: +SKIP
    [0.5, 0.5, 1.299, 1.299]
    """
    logging.debug('HIP: computing Gray STD for image %r and segm %r', np.shape(img), np.shape(seg))
    sess = _gray_stats_session(img, seg)
    _, _, var = sess.gray_stats(mean=False, energy=False, var=True)
    sess.close()
    return np.sqrt(var)


cython_img3d_gray_mean = hip_img3d_gray_mean
cython_img3d_gray_energy = hip_img3d_gray_energy
cython_img3d_gray_std = hip_img3d_gray_std


def numpy_img3d_gray_mean(img, seg):
    img, seg = np.asarray(img), np.asarray(seg)
    _check_gray_image_segm(img, seg)
    return _gray_sums(img, seg)


def numpy_img3d_gray_energy(img, seg):
    img, seg = np.asarray(img), np.asarray(seg)
    _check_gray_image_segm(img, seg)
    return _gray_sums(np.asarray(img, dtype=np.float64)**2, seg)


def numpy_img3d_gray_std(img, seg, means=None):
    img, seg = np.asarray(img), np.asarray(seg)
    _check_gray_image_segm(img, seg)
    if means is None:
        means = numpy_img3d_gray_mean(img, seg)
    nb = int(np.max(seg)) + 1
    if len(means) < nb:
        raise ValueError('number of means (%i) should be equal to number of labels (%i)' % (len(means), nb))
    d = np.asarray(img, dtype=np.float64) - np.asarray(means)[seg]
    return np.sqrt(_gray_sums(d**2, seg))


def numpy_img3d_gray_median(img, seg):
    img, seg = np.asarray(img), np.asarray(seg)
    _check_gray_image_segm(img, seg)
    nb = int(np.max(seg)) + 1
    return _segmented_median(np.asarray(img, dtype=np.float64).ravel(), seg.ravel(), nb)


def compute_image3d_gray_statistic(image, segm, feature_flags=NAMES_FEATURE_FLAGS, ch_name='gray', sess=None):
    """ statistics of a gray volume per supervoxel (reference descriptors.py:705-784)

    >>> image = np.zeros((2, 3, 8))
    >>> image[0, :, 2:6] = 1
    >>> image[1, :, 3:7] = 3
    >>> segm = np.array([[[0, 0, 0, 0, 1, 1, 1, 1]] * 3,
    ...                  [[2, 2, 2, 2, 5, 5, 5, 5]] * 3])
    >>> features, names = compute_image3d_gray_statistic(image, segm)  # doctest: +SKIP
    >>> np.round(features, 3).tolist()  # doctest: +SKIP
    [[0.5, 0.5, 0.5, 0.5, 0.25], [0.5, 0.5, 0.5, 0.5, -0.25], [0.75, 1.299, 2.25, 0.0, 0.75],
     [0.0, 0.0, 0.0, 0.0, 0.0], [0.0, 0.0, 0.0, 0.0, 0.0], [2.25, 1.299, 6.75, 3.0, -1.125]]
    >>> names  # doctest: +SKIP
    ['gray_mean', 'gray_std', 'gray_energy', 'gray_median', 'gray_meanGrad']
    """
    image = np.asarray(image)
    if sess is None:
        segm = np.asarray(segm)
    _check_gray_image_segm(image, segm)
    if not list(feature_flags):
        raise ValueError('some features has to be selected')
    if sess is None or 'median' in feature_flags or 'meanGrad' in feature_flags:
        image = np.nan_to_num(image)       # (a resident session already holds the caller-checked finite volume)
    features = []
    want = [f in feature_flags for f in ('mean', 'energy', 'std')]
    # median / meanGrad run on the device for the dtypes a session holds unchanged (the gradient volume keeps the dtype
    # of the source, descriptors.py:767-769: integer volumes truncate / wrap); exotic dtypes take the numpy route
    on_device = image.dtype in (np.uint8, np.float32, np.float64) and min(image.shape[1:]) >= 2
    extra = on_device and ('median' in feature_flags or 'meanGrad' in feature_flags)
    own = False
    if any(want) or extra:
        own = sess is None
        if own:
            sess = _hip.Volume3D(*segm.shape).upload(image).set_labels(segm)
    try:
        if any(want):
            mean, energy, var = sess.gray_stats(mean=want[0], energy=want[1], var=want[2])
            if want[0]:
                features.append(mean)
            if want[2]:
                features.append(np.sqrt(var))
            if want[1]:
                features.append(energy)
        if 'median' in feature_flags:
            features.append(sess.median() if on_device else numpy_img3d_gray_median(image, segm))
        if 'meanGrad' in feature_flags:
            if on_device:
                features.append(sess.mean_gradient())
            else:
                grad = np.zeros_like(image)
                for i in range(image.shape[0]):
                    grad[i] = np.sum(np.gradient(image[i]), axis=0)
                features.append(cython_img3d_gray_mean(grad, segm))
    finally:
        if own:
            sess.close()
    names = ['%s_%s' % (ch_name, n) for n in NAMES_FEATURE_FLAGS if n in feature_flags]
    _check_unrecognised_feature_names(feature_flags)
    features = np.nan_to_num(np.array(features)).T
    return _finished(features, names)


def _color_statistic_session(sess, image, segm, feature_flags, color_name):
    """colour statistics with the mean / std / energy columns taken from a device session that
    already holds ``image`` and ``segm`` (so nothing is re-uploaded)"""
    want = [f in feature_flags for f in ('mean', 'std', 'energy')]
    mean = energy = var = None
    if any(want):
        mean, energy, var = sess.color_stats(mean=want[0], energy=want[2], var=want[1])
    blocks = []
    if want[0]:
        blocks.append(mean)
    if want[1]:
        blocks.append(np.sqrt(var))
    if want[2]:
        blocks.append(energy)
    # median / meanGrad: on the device for the dtypes a session holds unchanged (the gradient image keeps the dtype of
    # the source, descriptors.py:842-844: integer images truncate / wrap); exotic dtypes take the numpy route
    on_device = np.asarray(image).dtype in (np.uint8, np.float32, np.float64) and min(np.shape(image)[:2]) >= 2
    if 'median' in feature_flags:
        blocks.append(sess.median() if on_device else numpy_img2d_color_median(image, segm))
    if 'meanGrad' in feature_flags:
        if on_device:
            blocks.append(sess.mean_gradient())
        else:
            grad = np.zeros_like(image)
            for i in range(3):
                grad[:, :, i] = np.sum(np.gradient(image[:, :, i]), axis=0)
            blocks.append(hip_img2d_color_mean(grad, segm))
    ch_names = ['%s-ch%i' % (color_name, i + 1) for i in range(3)]
    names = list(itertools.chain.from_iterable(['%s_%s' % (n, f) for n in ch_names] for f in NAMES_FEATURE_FLAGS
                                               if f in feature_flags))
    _check_unrecognised_feature_names(feature_flags)
    nb = sess.n_labels
    features = np.hstack(blocks) if blocks else np.empty((nb, 0))
    features = np.nan_to_num(features)
    return _finished(features, names)


def compute_image2d_color_statistic(image, segm, feature_flags=NAMES_FEATURE_FLAGS, color_name='color'):
    """ statistics of a colour image per superpixel, columns ordered mean, std, energy, median,
    meanGrad (reference descriptors.py:787-863)

    :param ndarray image: H x W x 3
    :param ndarray segm: segmentation H x W
    :param list(str) feature_flags: subset of NAMES_FEATURE_FLAGS
    :param str color_name: prefix of the feature names
    :return tuple(ndarray,list(str)): np.ndarray<nb_samples, nb_features>, names

    >>> image = np.zeros((2, 10, 3))
    >>> image[:, 2:6, 0] = 1
    >>> image[:, 3:7, 1] = 3
    >>> image[:, 4:9, 2] = 2
    >>> segm = np.array([[0, 0, 0, 0, 0, 1, 1, 1, 1, 1],
    ...                  [0, 0, 0, 0, 0, 1, 1, 1, 1, 1]])
    >>> features, names = compute_image2d_color_statistic(image, segm)  # doctest: +SKIP
    >>> features.shape  # doctest: +SKIP
    (2, 15)
    """
    image, segm = np.asarray(image), np.asarray(segm)
    _check_color_image(image)
    _check_color_image_segm(image, segm)
    image = np.nan_to_num(image)
    sess = _hip.Image2D(segm.shape[0], segm.shape[1]).upload(image).set_labels(segm)
    out = _color_statistic_session(sess, image, segm, feature_flags, color_name)
    sess.close()
    return out


def norm_features(features, scaler=None):
    """ standardise features (zero mean, unit variance); returns (features, scaler) """
    if not scaler:
        scaler = preprocessing.StandardScaler()
        scaler.fit(features)
    return scaler.transform(features), scaler


# ------------------------------------------------------------------------------------------------
# Leung-Malik texture bank (reference descriptors.py:880-1106) -- host scipy, as the reference
# ------------------------------------------------------------------------------------------------


def make_gaussian_filter1d(vals, sigma, order=0):
    if order > 2:
        raise ValueError("Only orders up to 2 are supported")
    response = np.exp(-vals**2 / (2. * sigma**2))
    if order == 1:
        response = -response * vals
    elif order == 2:
        response = response * (vals**2 - sigma**2)
    return response / np.abs(response).sum()


def make_edge_filter2d(sig, phase, points, sup):
    gx = make_gaussian_filter1d(points[0, :], sigma=3 * sig)
    gy = make_gaussian_filter1d(points[1, :], sigma=sig, order=phase)
    ft = (gx * gy).reshape(sup, sup)
    return ft / np.abs(ft).sum()


def create_filter_bank_lm_2d(radius=16, sigmas=DEFAULT_FILTERS_SIGMAS, nb_orient=8):
    """ Leung-Malik bank: per sigma rotated edge + bar batteries, Gaussian, LoG(sigma), LoG(sigma^2)

    >>> filters, names = create_filter_bank_lm_2d(6, SHORT_FILTERS_SIGMAS, 2)
    >>> [f.shape for f in filters][:5]
    [(2, 13, 13), (2, 13, 13), (1, 13, 13), (1, 13, 13), (1, 13, 13)]
    >>> names[:5]
    ['sigma1.4-edge', 'sigma1.4-bar', 'sigma1.4-Gauss', 'sigma1.4-GaussLap', 'sigma1.4-GaussLap2']
    """
    logging.debug('creating Leung-Malik filter bank')
    support = 2 * radius + 1
    x, y = np.mgrid[-radius:radius + 1, radius:-radius - 1:-1]
    org_pts = np.vstack([x.ravel(), y.ravel()])
    impulse = np.zeros((support, support))
    impulse[radius, radius] = 1
    filters, names = [], []
    for sigma in sigmas:
        edges, bars = [], []
        for orient in range(nb_orient):
            angle = np.pi * orient / nb_orient  # half circle: the filters are symmetric
            c, s = np.cos(angle), np.sin(angle)
            rot_points = np.dot(np.array([[c, -s], [s, c]]), org_pts)
            edges.append(make_edge_filter2d(sigma, 1, rot_points, support))
            bars.append(make_edge_filter2d(sigma, 2, rot_points, support))
        filters += [np.asarray(edges), np.asarray(bars)]
        filters.append(ndimage.gaussian_filter(impulse, sigma)[np.newaxis, :, :])
        filters.append(ndimage.gaussian_laplace(impulse, sigma)[np.newaxis, :, :])
        filters.append(ndimage.gaussian_laplace(impulse, sigma**2)[np.newaxis, :, :])
        names += ['sigma%.1f-%s' % (sigma, n) for n in ['edge', 'bar', 'Gauss', 'GaussLap', 'GaussLap2']]
    return filters, names


def compute_img_filter_response2d(img, filter_battery):
    """ responses of one battery on a 2D image; maximum over orientations for multi-kernel batteries """
    if filter_battery.ndim != 3:
        raise ValueError('wrong battery dim %r' % filter_battery.shape)
    responses = np.array([ndimage.convolve(img, fl) for fl in filter_battery])
    return np.max(responses, axis=0) if filter_battery.shape[0] > 1 else responses[0]


def compute_img_filter_response3d(img, filter_battery):
    """ slice-wise :func:`compute_img_filter_response2d` """
    logging.debug('compute image filter response in 3D')
    return np.array([compute_img_filter_response2d(img[i, :, :], filter_battery) for i in range(img.shape[0])])


def image_subtract_gauss_smooth(img, sigma):
    """ subtract a slice-wise Gaussian-smoothed copy (first axis independent) """
    if sigma <= 0:
        return img
    smooth = np.zeros(img.shape)
    for i in range(img.shape[0]):
        smooth[i, :, :] = ndimage.gaussian_filter(img[i, :, :].astype(float), sigma)
    return img - smooth


@functools.lru_cache(maxsize=4)
def _select_bank(bank_type):
    """the bank of a ``tLM`` / ``tLM_short`` group, built once per process (the arrays are shared: read-only for every caller;
    their identity is what :meth:`_hip.Image2D.lm_features` keys its factorisation on)"""
    if bank_type == 'short':
        filters, names = create_filter_bank_lm_2d(sigmas=SHORT_FILTERS_SIGMAS, nb_orient=4)
    else:
        filters, names = create_filter_bank_lm_2d()
    for battery in filters:
        battery.setflags(write=False)
    return filters, names


def _normalise_response(response):
    """clip and rescale one battery response globally, descriptors.py:1088-1094"""
    response[response > MAX_SIGNAL_RESPONSE] = MAX_SIGNAL_RESPONSE
    norm = np.sqrt(np.sum(response**2))
    if norm == 0 or abs(norm) == np.inf:
        return np.zeros(response.shape)
    return (response * (np.log(1 + norm) / 0.03)) / norm


def compute_texture_desc_lm_img3d_val(img, seg, feature_flags, bank_type='normal'):
    """ Leung-Malik texture statistics of a gray volume (reference descriptors.py:997-1038) """
    img, seg = np.asarray(img), np.asarray(seg)
    _check_gray_image_segm(img, seg)
    logging.debug('compute texture descriptors using Leung-Malik')
    filters, fl_names = _select_bank(bank_type)
    if set(feature_flags) <= {'mean', 'std', 'energy'} and all(len(f) <= 8 for f in filters):
        return _texture_desc_lm_device3d(img, seg, feature_flags, filters, fl_names)
    img = image_subtract_gauss_smooth(img, 150)
    features, names = [], []
    for battery, fl_name in zip(filters, fl_names):
        response = _normalise_response(compute_img_filter_response3d(img, battery))
        fts, n = compute_image3d_gray_statistic(response, seg, feature_flags, fl_name)
        features.append(fts)
        names += n
    features = np.nan_to_num(np.concatenate(tuple(features), axis=1))
    features[features == 0] = 0
    names = ['tLM_%s' % name for name in names]
    if features.shape[1] != len(names):
        raise ValueError('features: %r and names %r' % (features.shape, names))
    return features, names


def compute_texture_desc_lm_img2d_clr(img, seg, feature_flags, bank_type='normal'):
    """ Leung-Malik texture statistics of a colour image (reference descriptors.py:1041-1106)

    >>> h, w, step = 30, 20, 5
    >>> np.random.seed(0)
    >>> seg = (np.arange(h)[:, None] // step) * (w // step) + np.arange(w)[None, :] // step
    >>> img = np.random.random((h, w, 3))
    >>> features, names = compute_texture_desc_lm_img2d_clr(img, seg, ['mean', 'std', 'median'],
    ...                                                     bank_type='short')  # doctest: +SKIP
    >>> features.shape  # doctest: +SKIP
    (24, 135)
    """
    img, seg = np.asarray(img), np.asarray(seg)
    _check_color_image(img)
    logging.debug('compute texture descriptors using Leung-Malik')
    filters, fl_names = _select_bank(bank_type)
    if _texture_on_device(feature_flags, filters):
        return _texture_desc_lm_device(img, seg, feature_flags, filters, fl_names)
    # host path (median / meanGrad need the response on the host): scipy as the reference
    # scalar sigma on all three axes, channel axis included (descriptors.py:1078)
    img = img - ndimage.gaussian_filter(img.astype(float), 150)
    img_roll = np.rollaxis(img, -1, 0)
    sess = _hip.Image2D(seg.shape[0], seg.shape[1]).set_labels(seg)
    features, names = [], []
    for battery, fl_name in zip(filters, fl_names):
        response = np.rollaxis(_normalise_response(compute_img_filter_response3d(img_roll, battery)), 0, 3)
        response = np.nan_to_num(np.ascontiguousarray(response))
        sess.upload(response)
        fts, ns = _color_statistic_session(sess, response, seg, feature_flags, fl_name)
        features.append(fts)
        names += ns
    sess.close()
    return _finish_texture(features, names)


def _texture_on_device(feature_flags, filters):
    """the statistics the device forms of a filter response, and batteries it takes (at most 8 kernels)"""
    return set(feature_flags) <= {'mean', 'std', 'energy'} and all(len(f) <= 8 for f in filters)


def resident_feature_groups(feature_flags):
    """the descriptor groups of ``feature_flags`` in the column order of :func:`compute_selected_features_color2d` (the colour
    statistics, then the Leung-Malik ones) as ``[(kind, flags, batteries, columns)]`` when ALL of them can be formed and kept on
    the device -- 'color' in RGB and 'tLM' / 'tLM_<bank>' with mean / std / energy; otherwise None (converted colour spaces,
    median, meanGrad: the general path)"""
    order = [k for k in feature_flags if k.startswith('color')] + [k for k in feature_flags if k.startswith('tLM')]
    if not order or len(order) != len(feature_flags):
        return None
    groups = []
    for key in order:
        flags = set(feature_flags[key])
        if not flags or not flags <= {'mean', 'std', 'energy'}:
            return None
        if key == 'color':
            groups.append(('color', flags, None, 3 * len(flags)))
        elif key == 'tLM' or key.startswith('tLM_'):
            filters, _ = _select_bank(key.split('_')[-1] if '_' in key else 'normal')
            if not _texture_on_device(flags, filters) or len({np.shape(f)[1:] for f in filters}) != 1:
                return None
            groups.append(('tLM', flags, filters, 3 * len(flags) * len(filters)))
        else:
            return None
    return groups


def resident_feature_table(sess, groups):
    """forms the K x F table of ``groups`` (:func:`resident_feature_groups`) on the device session that holds the image and its
    label map -- nothing comes to the host, nothing is waited for; returns F"""
    total = sum(g[3] for g in groups)
    column, prepared = 0, False
    for kind, flags, filters, width in groups:
        sess.features_place(total, column)
        want = dict(mean='mean' in flags, std='std' in flags, energy='energy' in flags)
        if kind == 'color':
            sess.features_color(to_host=False, **want)
        else:
            if not prepared:
                sess.lm_prepare(150.)
                prepared = True
            sess.lm_features(filters, MAX_SIGNAL_RESPONSE, to_host=False, **want)
        column += width
    return total


def _finish_texture(features, names):
    features = np.nan_to_num(np.concatenate(tuple(features), axis=1))
    features[features == 0] = 0
    names = ['tLM_%s' % name for name in names]
    if features.shape[1] != len(names):
        raise ValueError('features: %r and names %r' % (features.shape, names))
    return features, names


def _texture_desc_lm_device(img, seg, feature_flags, filters, fl_names, sess=None):
    """Leung-Malik statistics entirely on the GPU: high-pass, filter batteries with orientation
    maximum, global norm, per-superpixel statistics (texture.hip + stats.hip)"""
    own = sess is None
    if own:
        sess = _hip.Image2D(seg.shape[0], seg.shape[1]).upload(np.nan_to_num(img)).set_labels(seg)
    sess.lm_prepare(150.)
    want = [f in feature_flags for f in ('mean', 'std', 'energy')]
    features, names = [], []
    one_call = getattr(sess, 'lm_features', None) if any(want) and len({np.shape(f)[1:] for f in filters}) == 1 else None
    if one_call is not None:
        # all batteries in one call: the norm of a battery never comes to the host (60 synchronisations per image less)
        features.append(one_call(filters, MAX_SIGNAL_RESPONSE, mean=want[0], std=want[1], energy=want[2]))
    for battery, fl_name in zip(filters, fl_names):
        if one_call is None:
            norm = sess.lm_battery(battery, MAX_SIGNAL_RESPONSE)
            nb = sess.n_labels
            if norm == 0 or abs(norm) == np.inf:
                mean = energy = var = np.zeros((nb, 3))
            else:
                mean, energy, var = sess.response_stats(np.log(1 + norm) / 0.03, norm, mean=want[0], energy=want[2],
                                                        var=want[1])
            blocks = []
            if want[0]:
                blocks.append(mean)
            if want[1]:
                blocks.append(np.sqrt(var))
            if want[2]:
                blocks.append(energy)
            fts = np.nan_to_num(np.hstack(blocks))
            fts[fts == 0] = 0
            features.append(fts)
        ch_names = ['%s-ch%i' % (fl_name, i + 1) for i in range(3)]
        names += list(itertools.chain.from_iterable(['%s_%s' % (n, f) for n in ch_names] for f in NAMES_FEATURE_FLAGS
                                                    if f in feature_flags))
    _check_unrecognised_feature_names(feature_flags)
    if own:
        sess.close()
    return _finish_texture(features, names)


def _texture_desc_lm_device3d(img, seg, feature_flags, filters, fl_names):
    """Leung-Malik statistics of a gray volume on the GPU: per-slice high-pass and filter batteries (the slices are the
    planes the 2-D kernels work on), one norm over the volume, per-supervoxel statistics"""
    sess = _hip.Volume3D(*seg.shape).upload(np.nan_to_num(img)).set_labels(seg)
    try:
        sess.lm_prepare(150.)
        want = [f in feature_flags for f in ('mean', 'std', 'energy')]
        features, names = [], []
        for battery, fl_name in zip(filters, fl_names):
            norm = sess.lm_battery(battery, MAX_SIGNAL_RESPONSE)
            nb = sess.n_labels
            if norm == 0 or abs(norm) == np.inf:
                mean = energy = var = np.zeros(nb)
            else:
                mean, energy, var = sess.response_stats(np.log(1 + norm) / 0.03, norm, mean=want[0], energy=want[2], var=want[1])
            blocks = ([mean] if want[0] else []) + ([np.sqrt(var)] if want[1] else []) + ([energy] if want[2] else [])
            fts = np.nan_to_num(np.array(blocks)).T
            fts[fts == 0] = 0
            features.append(fts)
            names += ['%s_%s' % (fl_name, f) for f in NAMES_FEATURE_FLAGS if f in feature_flags]
    finally:
        sess.close()
    _check_unrecognised_feature_names(feature_flags)
    features = np.nan_to_num(np.concatenate(tuple(features), axis=1))
    features[features == 0] = 0
    names = ['tLM_%s' % name for name in names]
    if features.shape[1] != len(names):
        raise ValueError('features: %r and names %r' % (features.shape, names))
    return features, names


# ------------------------------------------------------------------------------------------------
# feature-set selection (reference descriptors.py:1109-1285)
# ------------------------------------------------------------------------------------------------


def compute_selected_features_gray3d(img, segments, feature_flags=FEATURES_SET_COLOR, sess=None):
    """ selected features of a gray volume

    >>> np.random.seed(0)
    >>> img = np.random.random((2, 10, 15))
    >>> slic = np.zeros((2, 10, 15), dtype=int)
    >>> slic[:, :, :7] += 1
    >>> slic[1, :, :] += 2
    >>> fts, names = compute_selected_features_gray3d(img, slic, {'color': ('mean', 'std', 'median')})  # doctest: +SKIP
    >>> fts.shape  # doctest: +SKIP
    (4, 3)
    >>> names  # doctest: +SKIP
    ['gray_mean', 'gray_std', 'gray_median']
    """
    img = np.asarray(img)
    if sess is None:
        segments = np.asarray(segments)
    _check_gray_image_segm(img, segments)
    if not feature_flags:
        raise ValueError('some features has to be selected')
    features, names = [], []
    if any(k.startswith('color') for k in feature_flags):
        flags = np.unique([feature_flags[k] for k in feature_flags if k.startswith('color')])
        fts, ns = compute_image3d_gray_statistic(img, segments, flags, sess=sess)
        features.append(fts)
        names += ns
    for k in [k for k in feature_flags if k.startswith('tLM')]:
        bank_type = k.split('_')[-1] if '_' in k else 'normal'
        fts, ns = compute_texture_desc_lm_img3d_val(img, segments, feature_flags[k], bank_type)
        features.append(fts)
        names += ns
    _check_unrecognised_feature_group(feature_flags)
    if not features:
        logging.error('not supported features: %r', feature_flags)
    features = np.nan_to_num(np.concatenate(tuple(features), axis=1))
    return _finished(features, names)


def compute_selected_features_gray2d(img, segments, features_flags=FEATURES_SET_ALL):
    """ selected features of a gray 2D image (treated as a one-slice volume)

    >>> image = np.zeros((2, 10))
    >>> image[0, 2:6] = 1
    >>> image[1, 3:7] = 3
    >>> segm = np.array([[0, 0, 0, 0, 0, 1, 1, 1, 1, 1],
    ...                  [0, 0, 0, 0, 0, 1, 1, 1, 1, 1]])
    >>> features, names = compute_selected_features_gray2d(image, segm, {'color': ('mean', 'std', 'median')})  # doctest: +SKIP
    >>> np.round(features, 3).tolist()  # doctest: +SKIP
    [[0.9, 1.136, 0.5], [0.7, 1.187, 0.0]]
    """
    img, segments = np.asarray(img), np.asarray(segments)
    _check_gray_image_segm(img, segments)
    return compute_selected_features_gray3d(img[np.newaxis, ...], segments[np.newaxis, ...], features_flags)


def _convert_color(img, clr):
    from pyimsegm_amd.utilities.data_io import convert_img_color_from_rgb
    return convert_img_color_from_rgb(img, clr)


def _selected_features_color2d(img, segments, feature_flags, sess=None):
    _check_color_image(img)
    own = sess is None
    features, names = [], []
    for k in [k for k in feature_flags if k.startswith('color')]:
        clr = k.split('_')[-1] if '_' in k else 'rgb'
        if '_' in k:
            img_color = np.nan_to_num(_convert_color(img, clr))
            fts, ns = compute_image2d_color_statistic(img_color, segments, feature_flags[k], color_name=clr)
        else:
            if sess is None:
                sess = _hip.Image2D(segments.shape[0], segments.shape[1]).upload(np.nan_to_num(img)).set_labels(segments)
            fts, ns = _color_statistic_session(sess, img, segments, feature_flags[k], clr)
        features.append(fts)
        names += ns
    for k in [k for k in feature_flags if k.startswith('tLM')]:
        bank_type = k.split('_')[-1] if '_' in k else 'normal'
        filters, fl_names = _select_bank(bank_type)
        if _texture_on_device(feature_flags[k], filters):
            # on the session that holds the image and the labels already (one upload for all groups)
            _check_color_image(img)
            if sess is None:
                sess = _hip.Image2D(segments.shape[0], segments.shape[1]).upload(np.nan_to_num(img)).set_labels(segments)
            fts, ns = _texture_desc_lm_device(img, segments, feature_flags[k], filters, fl_names, sess=sess)
        else:
            fts, ns = compute_texture_desc_lm_img2d_clr(img, segments, feature_flags[k], bank_type)
        features.append(fts)
        names += ns
    if own and sess is not None:
        sess.close()
    _check_unrecognised_feature_group(feature_flags)
    if not features:
        logging.error('not supported features: %r', feature_flags)
        features = [np.empty((int(np.max(segments)) + 1, 0))]
    features = np.nan_to_num(np.concatenate(tuple(features), axis=1))
    return _finished(features, names)


def compute_selected_features_color2d(img, segments, feature_flags=FEATURES_SET_ALL):
    """ selected features of a colour 2D image (reference descriptors.py:1207-1270)

    >>> image = np.zeros((2, 10, 3))
    >>> image[:, 2:6, 0] = 1
    >>> image[:, 3:7, 1] = 3
    >>> image[:, 4:9, 2] = 2
    >>> segm = np.array([[0, 0, 0, 0, 0, 1, 1, 1, 1, 1],
    ...                  [0, 0, 0, 0, 0, 1, 1, 1, 1, 1]])
    >>> features, names = compute_selected_features_color2d(image, segm,
    ...                                   {'color': ('mean', 'std', 'median')})  # doctest: +SKIP
    >>> np.round(features, 3)  # doctest: +SKIP
    array([[0.6 , 1.2 , 0.4 , 0.49, 1.47, 0.8 , 1.  , 0.  , 0.  ],
           [0.2 , 1.2 , 1.6 , 0.4 , 1.47, 0.8 , 0.  , 0.  , 2.  ]])
    """
    return _selected_features_color2d(np.asarray(img), np.asarray(segments), feature_flags)


def compute_selected_features_img2d(image, segm, features_flags=FEATURES_SET_COLOR):
    """ dispatch on the image type: H x W x 3 colour or H x W gray """
    image, segm = np.asarray(image), np.asarray(segm)
    if image.ndim == 3 and image.shape[2] == 3:
        return compute_selected_features_color2d(image, segm, features_flags)
    if image.ndim == 2:
        return compute_selected_features_gray2d(image, segm, features_flags)
    logging.error('invalid image size - %r', image.shape)


# ------------------------------------------------------------------------------------------------
# label histograms around positions and ray features (reference descriptors.py:1371-1660); the
# natives behind them (features_cython.pyx:222-282) run batched on the device
# ------------------------------------------------------------------------------------------------


def adjust_bounding_box_crop(image_size, bbox_size, position):
    """ clip a box of ``bbox_size`` centred at ``position`` to the image (reference ``descriptors.py:1371-1409``)

    :return (), (), (), (): begin / end of the crop in the image, begin / end of the matching part of the box

    >>> adjust_bounding_box_crop((50, 50), (7, 7), (20, 20))
    ((17, 17), (24, 24), (0, 0), (7, 7))
    >>> adjust_bounding_box_crop((50, 50), (15, 15), (20, 45))
    ((13, 38), (28, 50), (0, 0), (15, 12))
    >>> adjust_bounding_box_crop((50, 50), (15, 15), (5, 5))
    ((0, 0), (13, 13), (2, 2), (15, 15))
    >>> adjust_bounding_box_crop((50, 50), (80, 80), (20, 20))
    ((0, 0), (50, 50), (20, 20), (70, 70))
    """
    if len(image_size) != len(bbox_size):
        raise ValueError('incompatible sizes %r != %r' % (image_size, bbox_size))
    im_begin, im_end, bb_begin, bb_end = [], [], [], []
    for size, box, pos in zip(image_size, bbox_size, position):
        size, box, pos = int(size), int(box), int(pos)
        half_lo, half_hi = box // 2, box - box // 2             # floor(box / 2), ceil(box / 2)
        lo, hi = max(pos - half_lo, 0), min(pos + half_hi, size)
        im_begin.append(lo)
        im_end.append(hi)
        bb_begin.append(half_lo - pos if lo == 0 else 0)
        bb_end.append(half_lo + (size - pos) if hi == size else box)
    if [e - b for b, e in zip(im_begin, im_end)] != [e - b for b, e in zip(bb_begin, bb_end)]:
        raise ValueError('different sizes of image %r and bounding box %r mask'
                         % ([e - b for b, e in zip(im_begin, im_end)], [e - b for b, e in zip(bb_begin, bb_end)]))
    return tuple(im_begin), tuple(im_end), tuple(bb_begin), tuple(bb_end)


def hip_label_hist_seg2d(segm_select, struc_elem, nb_labels):
    """ histogram of the labels under a structuring element (``computeLabelHistogram2d``) on the device

    >>> segm = np.zeros((10, 10), dtype=int)
    >>> segm[1:9, 2:8] = 1
    >>> segm[3:7, 4:6] = 2
    >>> hip_label_hist_seg2d(segm[2:5, 4:7], np.ones((3, 3)), 3).tolist()  # doctest: +SKIP
    [0.0, 5.0, 4.0]
    """
    segm_select, struc_elem = np.asarray(segm_select), np.asarray(struc_elem)
    if segm_select.shape != struc_elem.shape:
        raise ValueError('segm. %r and mask %r sizes do not match' % (segm_select.shape, struc_elem.shape))
    if segm_select.dtype.kind == 'f':
        segm_select = np.where(np.isnan(segm_select), -1, segm_select)      # NaN marks "no label" (descriptors.py:1490)
    h, w = segm_select.shape
    hist = _hip.label_hist2d(segm_select, [[0, 0, h, w, 0, 0]], struc_elem, nb_labels)[0]
    return np.array(hist, dtype=float)


cython_label_hist_seg2d = hip_label_hist_seg2d


def compute_label_hist_segm(segm, position, struc_elem, nb_labels):
    """ label histogram of the neighbourhood ``struc_elem`` around ``position`` and the size of that neighbourhood
    (reference ``descriptors.py:1411-1461``)

    >>> segm = np.zeros((10, 10), dtype=int)
    >>> segm[1:9, 2:8] = 1
    >>> segm[3:7, 4:6] = 2
    >>> compute_label_hist_segm(segm, [6, 6], np.ones((3, 3)), 3)  # doctest: +SKIP
    (array([ 0.,  7.,  2.]), 9.0)
    """
    hists, sizes = compute_label_hist_positions(segm, [position], struc_elem, nb_labels)
    return hists[0], sizes[0]


def compute_label_hist_positions(segm, positions, struc_elem, nb_labels):
    """ :func:`compute_label_hist_segm` for many positions in ONE launch (a wave per position)

    :return tuple(ndarray,ndarray): histograms P x nb_labels (float), sizes of the clipped neighbourhoods P
    """
    segm, struc_elem = np.asarray(segm), np.asarray(struc_elem)
    windows, sizes = [], []
    for position in positions:
        if segm.ndim != len(position):
            raise ValueError('dim of position %r should match the segmentation %r dim' % (position, segm.shape))
        position = [int(p) for p in position]
        im_begin, im_end, bb_begin, bb_end = adjust_bounding_box_crop(segm.shape, struc_elem.shape, position)
        windows.append([im_begin[0], im_begin[1], im_end[0] - im_begin[0], im_end[1] - im_begin[1], bb_begin[0], bb_begin[1]])
        sizes.append(np.sum(struc_elem[bb_begin[0]:bb_end[0], bb_begin[1]:bb_end[1]]))
    hists = _hip.label_hist2d(segm, windows, struc_elem, nb_labels)
    return np.array(hists, dtype=float), np.array(sizes, dtype=float)


def _ray_directions(angle_step):
    """per-angle step (d_row, d_col) of ``computeRayFeaturesBinary2d`` with the precision chain of features_cython.pyx:253-269:
    the float32 angle goes through ``np.deg2rad`` as a Python float (float64), the result is stored in a C ``float``; its
    sine / cosine are again formed in float64 and stored as ``float``; the division by the larger magnitude is float32"""
    angles = np.arange(0, 360, angle_step, dtype=np.float32)
    dirs = np.empty((len(angles), 2), dtype=np.float32)
    for i, ang in enumerate(angles):
        rad = np.float32(np.deg2rad(float(ang)))
        g0, g1 = np.float32(np.sin(float(rad))), np.float32(np.cos(float(rad)))
        gmax = max(abs(g0), abs(g1))
        dirs[i] = g0 / gmax, g1 / gmax
    return dirs


def hip_ray_features_positions(seg_binary, positions, angle_step=5., edge='up'):
    """ ray features of many positions in ONE launch (``computeRayFeaturesBinary2d`` per position): float32 P x A """
    edge_int = {'down': -1, 'up': 1}[edge]
    return _hip.ray_features_binary2d(np.asarray(seg_binary) != 0, positions, _ray_directions(float(angle_step)), edge_int)


def hip_ray_features_seg2d(seg_binary, position, angle_step=5., edge='up'):
    """ distances from ``position`` to the object boundary along rays (reference ``descriptors.py:1630-1659``)

    >>> seg_empty = np.zeros((100, 150), dtype=bool)
    >>> hip_ray_features_seg2d(seg_empty, (50, 75), 90).tolist()  # doctest: +SKIP
    [-1.0, -1.0, -1.0, -1.0]
    """
    return np.array(hip_ray_features_positions(seg_binary, [position], angle_step, edge)[0])


cython_ray_features_seg2d = hip_ray_features_seg2d


def compute_ray_features_segm_2d(seg_binary, position, angle_step=5., smooth_coef=0, edge='up'):
    """ ray features of one position, optionally smoothed along the angle (reference ``descriptors.py:1715-1758``:
    the ray casting followed by ``gaussian_filter1d``) """
    seg_binary = np.asarray(seg_binary)
    if seg_binary.ndim != len(position):
        raise ValueError('Segmentation dim of %r and position (%i) does not match' % (seg_binary.ndim, len(position)))
    ray_dist = hip_ray_features_seg2d(seg_binary.astype(bool), tuple(map(int, position)), angle_step, edge)
    if smooth_coef is not None and smooth_coef > 0:
        ray_dist = ndimage.gaussian_filter1d(ray_dist, smooth_coef)
    return ray_dist
