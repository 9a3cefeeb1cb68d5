"""A/B aid: bench.py against another build of the library (IMSEGM_HIP_LIBRARY), e.g. one from an older commit that lacks
the newest entry points -- signatures the library does not export are dropped before it is loaded.

    IMSEGM_HIP_LIBRARY=tools/ab/libimsegm_hip_old.so python tools/ab/run.py --no-other-configs --no-cpu-baseline --steps 60
"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from pyimsegm_amd import _hip  # noqa: E402

if os.environ.get('IMSEGM_HIP_LIBRARY'):
    probe = ctypes.CDLL(_hip.LIB_PATH)
    for name in list(_hip._SIGNATURES):
        if not hasattr(probe, name):
            del _hip._SIGNATURES[name]
import bench  # noqa: E402

bench.main()
