"""How fast does a pageable host array of 4.3 GB cross the host link -- one hipMemcpy, or slabs from several threads?"""
import ctypes as C
import sys
import threading
import time

import numpy as np

hip = C.CDLL('libamdhip64.so')
hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
H2D, D2H = 1, 2
n = int(sys.argv[1]) if len(sys.argv) > 1 else (1 << 30)
host = np.ones(n, dtype=np.float32)
back = np.empty(n, dtype=np.float32)
dev = C.c_void_p()
assert hip.hipMalloc(C.byref(dev), host.nbytes) == 0


def copy(kind, parts):
    step = (host.nbytes // parts + 4095) & ~4095
    def work(i):
        off = i * step
        size = min(step, host.nbytes - off)
        if size <= 0:
            return
        if kind == H2D:
            hip.hipSetDevice(0)
            hip.hipMemcpy(C.c_void_p(dev.value + off), C.c_void_p(host.ctypes.data + off), size, H2D)
        else:
            hip.hipSetDevice(0)
            hip.hipMemcpy(C.c_void_p(back.ctypes.data + off), C.c_void_p(dev.value + off), size, D2H)
    t0 = time.perf_counter()
    threads = [threading.Thread(target=work, args=(i, )) for i in range(parts)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    return host.nbytes / (time.perf_counter() - t0) / 1e9


for kind, name in ((H2D, 'H2D'), (D2H, 'D2H')):
    for parts in (1, 1, 2, 4, 8):
        print('%s %d thread(s): %.1f GB/s' % (name, parts, copy(kind, parts)), flush=True)
