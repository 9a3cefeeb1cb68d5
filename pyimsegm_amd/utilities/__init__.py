"""Small shared helpers of the package (error types used across the stage modules)."""


class ImageDimensionError(TypeError):
    """image / segmentation dimensions do not fit together

    Same role and base class as ``imsegm.utilities.ImageDimensionError``
    (reference ``imsegm/utilities/__init__.py:39``), so ``except TypeError`` keeps working.
    """
