"""Colour-space helpers used by the descriptor stage (numpy restatements of the few
``skimage.color`` conversions the reference reaches through
``imsegm/utilities/data_io.py:28-58``; scikit-image itself is not a dependency of this package).
Image file I/O of the reference (PNG / TIFF / NIfTI / ZVI readers) is out of scope."""
import numpy as np

_XYZ_FROM_RGB = np.array([[0.412453, 0.357580, 0.180423], [0.212671, 0.715160, 0.072169],
                          [0.019334, 0.119193, 0.950227]])
_XYZ_WHITE_D65 = np.array([0.95047, 1., 1.08883])


def _as_float_rgb(image):
    image = np.asarray(image)
    if image.dtype == np.uint8:
        return image[..., :3].astype(np.float64) / 255.
    return image[..., :3].astype(np.float64)


def rgb2hsv(rgb):
    """ RGB -> HSV, all channels in [0, 1] (same formulas as ``skimage.color.rgb2hsv``)
    """
    arr = _as_float_rgb(rgb)
    out = np.empty_like(arr)
    v = arr.max(-1)
    delta = arr.max(-1) - arr.min(-1)
    with np.errstate(invalid='ignore', divide='ignore'):
        s = delta / v
        s[delta == 0.] = 0.
        h = np.empty_like(v)
        idx = arr[..., 0] == v
        h[idx] = ((arr[..., 1] - arr[..., 2]) / delta)[idx]
        idx = arr[..., 1] == v
        h[idx] = 2. + ((arr[..., 2] - arr[..., 0]) / delta)[idx]
        idx = arr[..., 2] == v
        h[idx] = 4. + ((arr[..., 0] - arr[..., 1]) / delta)[idx]
        h = (h / 6.) % 1.
        h[delta == 0.] = 0.
    out[..., 0], out[..., 1], out[..., 2] = h, s, v
    out[np.isnan(out)] = 0
    return out


def rgb2xyz(rgb):
    arr = _as_float_rgb(rgb).copy()
    mask = arr > 0.04045
    arr[mask] = np.power((arr[mask] + 0.055) / 1.055, 2.4)
    arr[~mask] /= 12.92
    return arr @ _XYZ_FROM_RGB.T


def rgb2lab(rgb):
    arr = rgb2xyz(rgb) / _XYZ_WHITE_D65
    mask = arr > 0.008856
    arr[mask] = np.cbrt(arr[mask])
    arr[~mask] = 7.787 * arr[~mask] + 16. / 116.
    x, y, z = arr[..., 0], arr[..., 1], arr[..., 2]
    return np.stack([116. * y - 16., 500. * (x - y), 200. * (y - z)], axis=-1)


def rgb2luv(rgb):
    xyz = rgb2xyz(rgb)
    x, y, z = xyz[..., 0], xyz[..., 1], xyz[..., 2]
    eps = np.finfo(np.float64).eps
    L = y / _XYZ_WHITE_D65[1]
    mask = L > 0.008856
    L = np.where(mask, 116. * np.cbrt(np.where(mask, L, 1.)) - 16., 903.3 * L)
    u0 = 4 * _XYZ_WHITE_D65[0] / np.dot([1, 15, 3], _XYZ_WHITE_D65)
    v0 = 9 * _XYZ_WHITE_D65[1] / np.dot([1, 15, 3], _XYZ_WHITE_D65)
    denom = x + 15 * y + 3 * z + eps
    u = 13. * L * (4. * x / denom - u0)
    v = 13. * L * (9. * y / denom - v0)
    return np.stack([L, u, v], axis=-1)


_HED_FROM_RGB = np.linalg.inv(np.array([[0.65, 0.70, 0.29], [0.07, 0.99, 0.11], [0.27, 0.57, 0.78]]))


def rgb2hed(rgb):
    """ RGB -> Haematoxylin-Eosin-DAB (colour deconvolution as ``skimage.color.rgb2hed`` of scikit-image 0.18: the
    stain values are not clipped at zero; newer releases clip them) """
    arr = np.maximum(_as_float_rgb(rgb), 1e-6)
    return (np.log(arr) / np.log(1e-6)) @ _HED_FROM_RGB


#: conversion function from RGB color space
DICT_CONVERT_COLOR_FROM_RGB = {'hsv': rgb2hsv, 'luv': rgb2luv, 'lab': rgb2lab, 'hed': rgb2hed, 'xyz': rgb2xyz}


def convert_img_color_from_rgb(image, color_space):
    """ convert image colour space from RGB to ``color_space`` (unknown names: image returned as is)
    """
    image = np.asarray(image)
    if image.ndim == 3 and image.shape[-1] in (3, 4) and color_space in DICT_CONVERT_COLOR_FROM_RGB:
        image = DICT_CONVERT_COLOR_FROM_RGB[color_space](image)
    return image


def get_image2d_boundary_color(image, size=1):
    """ the dominant value on the image border of width ``size`` (reference ``utilities/data_io.py:1002-1036``):
    the most frequent label of a 2D (label) image, the per-channel median of a colour image
    """
    import logging
    image = np.asarray(image)
    size = int(size)
    strips = [image[:size, :], image[:, :size], image[-size:, :], image[:, -size:]]
    if image.ndim == 2:
        pixels = np.concatenate([strip.ravel() for strip in strips])
        colour = np.argmax(np.bincount(pixels))
    elif image.ndim == 3:
        pixels = np.concatenate([strip.reshape(-1, image.shape[-1]) for strip in strips], axis=0)
        colour = np.median(pixels, axis=0)
    else:
        logging.error('not supported image dim: %r', image.shape)
        colour = np.array(0)
    return np.asarray(colour).astype(image.dtype)
