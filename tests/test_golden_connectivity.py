"""``oracle.enforce_connectivity`` against the REAL scikit-image 0.18.3 (``_enforce_label_connectivity_cython``) on crafted label
maps -- ``tests/golden/connectivity.npz``, written by ``tests/golden/make_golden_connectivity.py`` under the build container's
conda Python 3.9.  Bit for bit, both start labels, 2-D and one volume."""
import importlib.util
import os
import zlib

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
_spec = importlib.util.spec_from_file_location('make_golden_connectivity', os.path.join(HERE, 'golden', 'make_golden_connectivity.py'))
GEN = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(GEN)


@pytest.fixture(scope='module')
def golden():
    return np.load(os.path.join(HERE, 'golden', 'connectivity.npz'))


@pytest.mark.parametrize('name', sorted(GEN.CASES))
@pytest.mark.parametrize('start_label', [0, 1])
def test_oracle_connectivity_equals_scikit_image(oracle, golden, name, start_label):
    assert str(golden['skimage_version']) == '0.18.3'
    _, min_size, max_size = GEN.CASES[name]
    lab = GEN.make(name) + start_label
    key = '%s_start%d' % (name, start_label)
    assert zlib.crc32(np.ascontiguousarray(lab, dtype=np.int32).tobytes()) == int(golden[key + '_input_crc']), 'generator drifted'
    got = oracle.enforce_connectivity(lab, min_size, max_size, start_label)
    assert np.array_equal(got, golden[key])
