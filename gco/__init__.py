"""Shim of the ``gco`` module (gco-wrapper / pyGCO) on top of the MI355X alpha-expansion kernel.

The reference binds GraphCut through ``from gco import cut_general_graph, cut_grid_graph`` (``imsegm/graph_cuts.py:17``,
``imsegm/region_growing.py:20``).  With this directory on ``sys.path`` -- it sits next to the ``imsegm`` overlay package --
those imports resolve to the device implementation, so the reference's own ``region_growing`` module (object
segmentation on pixels and on superpixels, the region-growing energy) runs its graph cuts on the GPU unchanged.  A real
gco-wrapper installed earlier on ``sys.path`` is NOT shadowed.  Only ``algorithm='expansion'`` exists on the device."""
from pyimsegm_amd._hip import cut_general_graph, cut_grid_graph  # noqa: F401

__all__ = ['cut_general_graph', 'cut_grid_graph']
