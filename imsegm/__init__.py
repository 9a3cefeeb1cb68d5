"""Alias package: ``import imsegm.<stage>`` resolves to the MI355X-native implementation in
``pyimsegm_amd`` -- the very same module objects, so attribute writes such as the driver's
``imsegm.descriptors.USE_CYTHON = False`` (reference ``run_segm_slic_model_graphcut.py:59``) reach the
real module.  Only the hot-path modules exist here."""
import importlib
import sys

import pyimsegm_amd

__version__ = pyimsegm_amd.__version__

for _name in ('utilities', 'utilities.data_io', 'superpixels', 'descriptors', 'graph_cuts', 'labeling', 'classification', 'pipelines'):
    _mod = importlib.import_module('pyimsegm_amd.' + _name)
    sys.modules['imsegm.' + _name] = _mod
    if '.' not in _name:
        globals()[_name] = _mod
del _name, _mod
