"""CPU-side checks of the product library: it loads, exports every symbol include/imsegm_hip.h
declares, and fails loudly (no CPU fallback) when no GPU is present."""
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def hip():
    from pyimsegm_amd import _hip, build
    build.build(verbose=False)
    return _hip


def test_exports_every_declared_symbol(hip):
    header = open(os.path.join(ROOT, 'include', 'imsegm_hip.h')).read()
    declared = set(re.findall(r'IMSEGM_API\s+[\w\s\*]*?\b(imsegm_\w+)\s*\(', header))
    assert declared, 'no declarations parsed'
    assert declared == set(hip.EXPORTED_SYMBOLS)
    lib = hip.load_library()
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.imsegm_version() >= 100


def test_no_silent_cpu_fallback(hip):
    if hip.device_count() > 0:
        pytest.skip('a GPU is present')
    with pytest.raises(hip.HipUnavailableError):
        hip.default_context()
    with pytest.raises(hip.HipUnavailableError):
        hip.Image2D(8, 8)


def test_gaussian_taps_match_scipy(hip, oracle):
    for sigma in (1.0, 0.2, 0.5, 2.5, 1. / 12):
        a, b = hip.gaussian_taps(sigma), oracle.gaussian_taps(sigma)
        assert np.array_equal(a, b)
    assert hip.gaussian_taps(0.) is None
