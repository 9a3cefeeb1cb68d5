"""How the host's mixture fit of config 5 (298 116 x 3 rows, GaussianMixture(4, full, n_init=9)) behaves when several run at once:
in threads of one interpreter (what volumes in flight do) against separate processes.

    python tools/fit_concurrency.py [rows]
"""
import os
import sys
import time
from concurrent.futures import ProcessPoolExecutor, ThreadPoolExecutor

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def table(rows, seed):
    rng = np.random.default_rng(seed)
    centres = np.array([[0.1, 0.02, 0.01], [0.35, 0.05, 0.13], [0.65, 0.05, 0.43], [0.95, 0.06, 0.9]])
    which = rng.integers(0, 4, rows)
    return centres[which] + 0.03 * rng.standard_normal((rows, 3))


def one_fit(args):
    rows, seed = args
    from pyimsegm_amd import graph_cuts as G
    feats = table(rows, seed)
    np.random.seed(seed)
    t0 = time.perf_counter()
    model = G.estim_class_model(feats, 4)
    return time.perf_counter() - t0, float(np.asarray(G.predict_proba(model, feats[:16])).sum())


def main():
    rows = int(sys.argv[1]) if len(sys.argv) > 1 else 298116
    print('cpus', os.cpu_count())
    one_fit((rows, 0))                                   # imports, pools
    alone = [one_fit((rows, 1))[0] for _ in range(3)]
    print('alone              %s' % ' '.join('%.3f' % t for t in alone))
    for n in (2, 3, 4):
        with ThreadPoolExecutor(n) as pool:
            t0 = time.perf_counter()
            times = [r[0] for r in pool.map(one_fit, [(rows, 10 + i) for i in range(n)])]
            wall = time.perf_counter() - t0
        print('%d threads          wall %.3f  each %s  -> %.3f s per fit' % (n, wall, ' '.join('%.3f' % t for t in times), wall / n))
    import multiprocessing as mp
    for n in (2, 3, 4):
        with ProcessPoolExecutor(n, mp_context=mp.get_context('spawn')) as pool:
            list(pool.map(one_fit, [(rows, 0)] * n))       # start-up
            t0 = time.perf_counter()
            times = [r[0] for r in pool.map(one_fit, [(rows, 10 + i) for i in range(n)])]
            wall = time.perf_counter() - t0
        print('%d processes        wall %.3f  each %s  -> %.3f s per fit' % (n, wall, ' '.join('%.3f' % t for t in times), wall / n))


if __name__ == '__main__':
    main()
