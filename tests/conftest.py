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


def _reload_library_env():
    """the library reads its IMSEGM_* debug switches once; tests that flip them tell it to read them again"""
    try:
        from pyimsegm_amd import _hip
        if _hip._lib is not None:
            _hip.reload_env()
    except Exception:
        pass


@pytest.fixture(autouse=True)
def _library_env_follows_the_test():
    _reload_library_env()          # (whatever the previous test's monkeypatch has undone)
    yield


@pytest.fixture
def monkeypatch(monkeypatch):
    """pytest's monkeypatch whose setenv / delenv also make libimsegm_hip.so read the environment again"""
    setenv, delenv = monkeypatch.setenv, monkeypatch.delenv

    def setenv_and_reload(name, value, prepend=None):
        setenv(name, value, prepend)
        _reload_library_env()

    def delenv_and_reload(name, raising=True):
        delenv(name, raising)
        _reload_library_env()

    monkeypatch.setenv, monkeypatch.delenv = setenv_and_reload, delenv_and_reload
    yield monkeypatch
