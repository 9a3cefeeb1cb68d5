"""Overlay package: ``import imsegm.<module>`` resolves

* to the MI355X-native implementation in ``pyimsegm_amd`` for the modules of the SLIC -> descriptors ->
  GraphCut hot path (``superpixels``, ``descriptors``, ``graph_cuts``, ``pipelines``, plus ``labeling`` and
  ``classification`` of the supervised path) -- the very same module objects, so attribute writes such as the
  driver's ``imsegm.descriptors.USE_CYTHON = False`` (reference ``run_segm_slic_model_graphcut.py:59``) reach
  the real module;
* to the reference's OWN files for everything else, when a reference ``imsegm`` package is installed next to
  this one (anywhere on ``sys.path`` behind this directory, or named by the environment variable
  ``IMSEGM_REFERENCE`` = the directory that holds ``imsegm/``): its directory is appended to ``__path__``, so
  ``imsegm.utilities.{data_io,drawing,experiments,...}``, ``imsegm.annotation``, ``imsegm.region_growing``,
  ``imsegm.ellipse_fitting`` import as they always did, and the reference's experiment drivers run unchanged;
* a name that a shadowed module of this repo does not define (e.g. ``imsegm.descriptors.compute_ray_features_segm_2d``,
  needed by ``region_growing``) falls back to the reference's module of the same name, loaded privately as
  ``imsegm._reference.<module>``.

Without an installed reference only the hot-path modules exist."""
import importlib
import importlib.util
import os
import sys
import types

import pyimsegm_amd

__version__ = pyimsegm_amd.__version__

#: modules this repo owns (the hot path); everything else belongs to the reference
SHADOWED = ('superpixels', 'descriptors', 'graph_cuts', 'labeling', 'classification', 'pipelines')
_HERE = os.path.dirname(os.path.abspath(__file__))


def _find_reference():
    """directory of an installed reference ``imsegm`` package (not this one), or None"""
    cands = []
    env = os.environ.get('IMSEGM_REFERENCE', '')
    if env:
        cands.append(env)
    cands += [p or '.' for p in sys.path]
    for base in cands:
        try:
            pkg = os.path.join(os.path.abspath(base), 'imsegm')
        except (TypeError, ValueError):
            continue
        if not os.path.isdir(pkg) or os.path.samefile(pkg, _HERE):
            continue
        # the reference is recognised by a stage module this overlay shadows plus its utilities package
        if os.path.isfile(os.path.join(pkg, '__init__.py')) and os.path.isfile(os.path.join(pkg, 'pipelines.py')) \
                and os.path.isdir(os.path.join(pkg, 'utilities')):
            return pkg
    return None


#: path of the reference package this overlay completes (None: stand-alone)
REFERENCE_PATH = None if os.environ.get('IMSEGM_REFERENCE', None) == '' else _find_reference()


def _reference_module(name):
    """the reference's own ``imsegm/<name>.py`` as a private module (never registered as ``imsegm.<name>``)"""
    full = 'imsegm._reference.' + name
    mod = sys.modules.get(full)
    if mod is None:
        if REFERENCE_PATH is None:
            raise ImportError('no reference imsegm package is installed')
        path = os.path.join(REFERENCE_PATH, name + '.py')
        spec = importlib.util.spec_from_file_location(full, path)
        if spec is None or not os.path.isfile(path):
            raise ImportError('the reference has no module %r' % name)
        mod = importlib.util.module_from_spec(spec)
        sys.modules[full] = mod
        try:
            spec.loader.exec_module(mod)
        except BaseException:
            sys.modules.pop(full, None)
            raise
    return mod


def _fallback_getattr(name):
    def __getattr__(attr):
        if attr.startswith('__') or REFERENCE_PATH is None:
            raise AttributeError('module %r has no attribute %r' % ('imsegm.' + name, attr))
        try:
            ref = _reference_module(name)
        except ImportError as ex:
            raise AttributeError('module %r has no attribute %r (reference fallback failed: %s)'
                                 % ('imsegm.' + name, attr, ex))
        try:
            return getattr(ref, attr)
        except AttributeError:
            raise AttributeError('module %r has no attribute %r' % ('imsegm.' + name, attr))
    return __getattr__


if REFERENCE_PATH is not None:
    __path__.append(REFERENCE_PATH)
    sys.modules.setdefault('imsegm._reference', types.ModuleType('imsegm._reference'))
    # the reference's utilities package (data_io, drawing, experiments, ...) as it is; its error type becomes
    # the one the HIP path raises, before any other reference module imports it by name
    import pyimsegm_amd.utilities as _own_util
    _util = importlib.import_module('imsegm.utilities')
    _util.ImageDimensionError = _own_util.ImageDimensionError
    del _util, _own_util
else:
    for _name in ('utilities', 'utilities.data_io'):
        _mod = importlib.import_module('pyimsegm_amd.' + _name)
        sys.modules['imsegm.' + _name] = _mod
        if '.' not in _name:
            globals()[_name] = _mod
    del _name, _mod

for _name in SHADOWED:
    _mod = importlib.import_module('pyimsegm_amd.' + _name)
    sys.modules['imsegm.' + _name] = _mod
    globals()[_name] = _mod
    if REFERENCE_PATH is not None and not hasattr(_mod, '__getattr__'):
        _mod.__getattr__ = _fallback_getattr(_name)
del _name, _mod
