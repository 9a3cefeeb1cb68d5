"""The crafted label maps of ``tests/test_gpu_connectivity.py`` (reduced in size) through the oracle and through the literal
pure-Python restatement of ``_enforce_label_connectivity_cython`` (``tests/test_oracle_slic.py``): the GPU tests compare with
the oracle, so the oracle has to be right on exactly these shapes -- combs with wide BFS frontiers, thin diagonals, salt noise,
oversize components, one row / one column -- and not only on SLIC label maps."""
import numpy as np
import pytest

from test_oracle_slic import literal_connectivity


def _blocks(h, w, bh, bw):
    yy, xx = np.mgrid[0:h, 0:w]
    return ((yy // bh) * ((w + bw - 1) // bw) + xx // bw).astype(np.int32)


def _salted(h, w, bh, bw, frac, seed):
    rng = np.random.RandomState(seed)
    lab = _blocks(h, w, bh, bw)
    m = rng.rand(h, w) < frac
    lab[m] = rng.randint(0, lab.max() + 1, m.sum())
    return lab


def _comb(h, w, teeth):
    lab = np.zeros((h, w), np.int32)
    lab[:, w // 2:] = 1
    lab[2, 2:2 + 2 * teeth] = 2
    for x in range(2, 2 + 2 * teeth, 2):
        lab[3:7, x] = 2
    return lab


def _diagonal(h, w, n):
    lab = _blocks(h, w, 16, 16)
    for i in range(n):
        lab[3 + i, 3 + i] = 99
        lab[3 + i, 4 + i] = 99
    return lab


CASES = [
    ('blocks', lambda: _blocks(23, 37, 5, 7), 10, 80),
    ('salt', lambda: _salted(40, 48, 8, 8, 0.05, 0), 20, 300),
    ('noise', lambda: np.random.RandomState(2).randint(0, 5, (30, 34)).astype(np.int32), 6, 60),
    ('comb', lambda: _comb(12, 60, 20), 200, 10000),
    ('diagonal', lambda: _diagonal(40, 40, 30), 100, 10000),
    ('oversize', lambda: _blocks(24, 24, 12, 12), 5, 50),
    ('one_row', lambda: _blocks(1, 60, 1, 7), 5, 30),
    ('one_column', lambda: _blocks(60, 1, 9, 1), 5, 30),
    ('volume', lambda: (np.arange(4)[:, None, None] // 2 * 10 + _blocks(10, 12, 5, 4)[None]).astype(np.int32), 8, 200),
]


@pytest.mark.parametrize('name,make,min_size,max_size', CASES, ids=[c[0] for c in CASES])
@pytest.mark.parametrize('start_label', [0, 1])
def test_oracle_connectivity_equals_literal_restatement(oracle, name, make, min_size, max_size, start_label):
    lab = make() + start_label
    lab3 = lab[None] if lab.ndim == 2 else lab
    want = literal_connectivity(lab3, min_size, max_size, start_label).reshape(lab.shape)
    got = oracle.enforce_connectivity(lab, min_size, max_size, start_label)
    assert np.array_equal(got, want)
