#!/usr/bin/env python
"""Golden vector of the `slic_mean` debug image (imsegm/pipelines.py:93: `skimage.color.label2rgb(slic, image, kind='avg')`)
from the REAL scikit-image 0.18.3 (build container: /opt/conda/bin/python3.9 tests/golden/make_golden_label2rgb.py), on the
superpixels of the real `skimage.segmentation.slic` of two images of tests/golden/skimage.npz (uint8 and float input).
In 0.18 `label2rgb(kind='avg')` runs with bg_label = -1: label 0 gets its mean colour like every other label."""
import os
import sys
import warnings
import zlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
warnings.filterwarnings('ignore')


def main():
    import skimage
    from skimage import color
    from make_golden_skimage import CASES_2D, make_input
    assert skimage.__version__.startswith('0.18'), skimage.__version__
    ref = np.load(os.path.join(HERE, 'skimage.npz'))
    out = {'skimage_version': np.array(skimage.__version__)}
    for name in ('voronoi', 'float'):
        img = make_input(CASES_2D[name][0])
        assert zlib.crc32(np.ascontiguousarray(img).tobytes()) == int(ref[name + '_crc'])
        slic = ref[name + '_final'].astype(np.int64)
        out[name + '_avg'] = np.asarray(color.label2rgb(slic, img, kind='avg'))
        print(name, out[name + '_avg'].dtype, out[name + '_avg'].shape, float(out[name + '_avg'].max()))
    np.savez_compressed(os.path.join(HERE, 'label2rgb.npz'), **out)


if __name__ == '__main__':
    sys.path.insert(0, HERE)
    main()
