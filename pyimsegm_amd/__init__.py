"""pyimsegm_amd -- MI355X-native SLIC -> superpixel descriptors -> GraphCut hot path of pyImSegm.

The public surface mirrors the reference's stage modules (``imsegm.superpixels``,
``imsegm.descriptors``, ``imsegm.graph_cuts``, ``imsegm.pipelines``); all heavy lifting happens in
hand-written HIP kernels (``csrc/``) reached through the C ABI of ``libimsegm_hip.so``.
Importing the package does not touch the GPU.
"""
__version__ = '0.1.0'
