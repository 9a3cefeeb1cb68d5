"""Small shared helpers of the package (error types used across the stage modules)."""


class ImageDimensionError(TypeError):
    """image / segmentation dimensions do not fit together

    Same role and base class as ``imsegm.utilities.ImageDimensionError``
    (reference ``imsegm/utilities/__init__.py:39``), so ``except TypeError`` keeps working.
    """


def reference_attribute(module, name):
    """attribute ``name`` of the REFERENCE's own ``imsegm/<module>.py`` -- for the names of a hot-path module that this
    package does not restate (they never touch the device): resolved when a reference package is installed behind the
    ``imsegm`` overlay (see ``imsegm/__init__.py``), an ``AttributeError`` that says so otherwise."""
    if name.startswith('__'):
        raise AttributeError(name)
    import importlib
    try:
        overlay = importlib.import_module('imsegm')
        owner = overlay._reference_module(module)
    except Exception as ex:
        raise AttributeError('module %r has no attribute %r -- that name is not part of the SLIC -> descriptors -> GraphCut path '
                             'and lives in the reference package, which is not installed here (%s)'
                             % ('pyimsegm_amd.' + module, name, ex))
    try:
        return getattr(owner, name)
    except AttributeError:
        raise AttributeError('module %r has no attribute %r' % ('pyimsegm_amd.' + module, name))
