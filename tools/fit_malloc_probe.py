"""does glibc's mmap threshold cost the host-side mixture fit?  Every numpy temporary of the EM loop of config 5 (298 116 rows: 2.4 -
7 MB each) is above the default threshold: mmap + page faults + munmap per operation.  With M_MMAP_THRESHOLD / M_TRIM_THRESHOLD
raised the blocks come from the heap and are recycled.  Same arithmetic, same bits -- only where the bytes live.

    python tools/fit_malloc_probe.py            # fit as it is, then with the thresholds raised (same process, in that order)
"""
import ctypes
import os
import sys
import time
import warnings

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
warnings.filterwarnings('ignore')
import numpy as np
from sklearn import mixture
from threadpoolctl import threadpool_limits

from pyimsegm_amd import graph_cuts as g

rng = np.random.default_rng(0)
K = 298116
X = np.vstack([rng.normal([0, 0, 0], [1, .5, .8], (K // 3, 3)), rng.normal([3, 1, 2], [.7, .6, .5], (K // 3, 3)),
               rng.normal([-2, 2, 1], [.5, .9, .6], (K - 2 * (K // 3), 3))])
X += rng.normal(0, 1.5, X.shape)


def fit():
    np.random.seed(3)
    t = time.perf_counter()
    m = g.fit_mixture_restarts(mixture.GaussianMixture(3, covariance_type='full', n_init=9, max_iter=99), X)
    return time.perf_counter() - t, m


limiter = threadpool_limits(limits=min(32, os.cpu_count() or 1))
base = [fit() for _ in range(4)]
print('default malloc      : %s s' % ' '.join('%.3f' % t for t, _ in base))
libc = ctypes.CDLL('libc.so.6')
M_TRIM_THRESHOLD, M_MMAP_THRESHOLD, M_TOP_PAD = -1, -3, -2
print('mallopt', libc.mallopt(M_MMAP_THRESHOLD, 1 << 30), libc.mallopt(M_TRIM_THRESHOLD, 1 << 31 - 1), libc.mallopt(M_TOP_PAD, 256 << 20))
tuned = [fit() for _ in range(4)]
print('thresholds raised   : %s s' % ' '.join('%.3f' % t for t, _ in tuned))
a, b = base[-1][1], tuned[-1][1]
print('same parameters:', all(np.array_equal(getattr(a, n), getattr(b, n)) for n in ('weights_', 'means_', 'covariances_', 'precisions_cholesky_')),
      'iterations', a.n_iter_, b.n_iter_)
