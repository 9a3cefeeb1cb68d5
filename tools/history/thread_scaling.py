"""Do worker threads scale on the pure C-ABI calls (GIL released), i.e. is the HIP runtime or the GIL the limit?"""
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyimsegm_amd import _hip  # noqa: E402
from pyimsegm_amd.superpixels import _open_session, _slic_params  # noqa: E402
from pyimsegm_amd.utilities.synthetic import voronoi_image  # noqa: E402

img = voronoi_image(2048, 2048)
n_seg, compact = _slic_params(img.shape[:2], 46, 0.2)


def run(nthreads, iters=20):
    ready = threading.Barrier(nthreads + 1)
    done = threading.Barrier(nthreads + 1)

    def worker():
        sess, mode = _open_session(img)
        sess.slic(n_seg, compact, sigma=1., normalize=mode)
        ready.wait()
        for _ in range(iters):
            sess.slic(n_seg, compact, sigma=1., normalize=mode)       # one C call, GIL released
            sess.color_stats()
            sess.graph()
        done.wait()

    ts = [threading.Thread(target=worker, daemon=True) for _ in range(nthreads)]
    for t in ts:
        t.start()
    ready.wait()
    t0 = time.perf_counter()
    done.wait()
    dt = time.perf_counter() - t0
    print('%d thread(s): %.3f ms per image (slic + stats + graph only)' % (nthreads, dt / (iters * nthreads) * 1e3))


for n in (1, 2, 3, 4):
    run(n)
