"""pytest configuration: markers, import path, oracle build."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: test needs a real MI355X (run with -m gpu)')


@pytest.fixture(scope='session')
def oracle():
    """the CPU oracle (test infrastructure); built on demand"""
    from oracle import oracle as orc
    orc.lib()
    return orc


@pytest.fixture(scope='session')
def ref_cython(oracle):
    """the reference's own features_cython.pyx compiled into oracle/_ref (None if not built)"""
    return oracle.ref_features_cython()
