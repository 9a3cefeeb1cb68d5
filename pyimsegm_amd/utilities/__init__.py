"""Small shared helpers of the package (error types used across the stage modules)."""


class ImageDimensionError(TypeError):
    """image / segmentation dimensions do not fit together

    Same role and base class as ``imsegm.utilities.ImageDimensionError``
    (reference ``imsegm/utilities/__init__.py:39``), so ``except TypeError`` keeps working.
    """


def reference_attribute(module, name):
    """attribute ``name`` of the REFERENCE's own ``imsegm/<module>.py`` -- for the names of a hot-path module that this
    package does not restate (they never touch the device).  Resolved only when the ``imsegm`` overlay of this repo has
    ALREADY been imported by the caller and found a reference package behind it (see ``imsegm/__init__.py``); an
    ``AttributeError`` that says so otherwise.  A probe (``hasattr``, ``getattr(..., None)``) imports nothing: the overlay
    is looked up in ``sys.modules``, never loaded from here."""
    if name.startswith('__'):
        raise AttributeError(name)
    import sys
    overlay = sys.modules.get('imsegm')
    finder = getattr(overlay, '_reference_module', None)
    if finder is None or getattr(overlay, 'REFERENCE_PATH', None) is None:
        raise AttributeError('module %r has no attribute %r -- that name is not part of the SLIC -> descriptors -> GraphCut path; '
                             'it lives in the reference package (`import imsegm` with a reference installed behind the overlay '
                             'makes it available here)' % ('pyimsegm_amd.' + module, name))
    try:
        owner = finder(module)
    except Exception as ex:
        raise AttributeError('module %r has no attribute %r (the reference module did not load: %s)'
                             % ('pyimsegm_amd.' + module, name, ex))
    try:
        return getattr(owner, name)
    except AttributeError:
        raise AttributeError('module %r has no attribute %r' % ('pyimsegm_amd.' + module, name))
