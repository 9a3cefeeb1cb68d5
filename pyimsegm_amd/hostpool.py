"""Helper processes for the host-side (scikit-learn / numpy) stages of the pipeline.

Between the kernels of one image the host evaluates the class model and the graph-cut terms
(``graph_cuts.predict_proba``, ``compute_unary_cost``, ``compute_pairwise_cost``, the edge weights) -- about a
millisecond of numpy per 2048 x 2048 image, all of it under the interpreter lock.  With several images in
flight on one GPU (worker threads, ``pipelines.NB_WORKERS``) that lock becomes the bottleneck.  The reference
spreads images over a pool of worker *processes* (``imsegm/utilities/experiments.py:392-403``); here the GPU
submission stays in threads of one process (one HIP context, zero-copy gathers) and only the numpy work moves
to helper processes, one round trip per image:

    features, edges, centres  ->  proba, unary cost, pairwise cost, edge weights

The helpers run the very same functions of ``graph_cuts`` (single-threaded BLAS), so the numbers are identical to
the in-process path.  They never touch the GPU.  Protocol: length-prefixed pickles over the helper's stdin /
stdout pipes.
"""
import os
import pickle
import queue
import struct
import subprocess
import sys
import threading


def _send(stream, obj):
    data = pickle.dumps(obj, protocol=pickle.HIGHEST_PROTOCOL)
    stream.write(struct.pack('<Q', len(data)))
    stream.write(data)
    stream.flush()


def _recv(stream):
    head = stream.read(8)
    if len(head) < 8:
        raise EOFError('host helper closed the pipe')
    size, = struct.unpack('<Q', head)
    data = stream.read(size)
    if len(data) < size:
        raise EOFError('host helper closed the pipe')
    return pickle.loads(data)


def graph_cut_terms(model, features, edges, centres, gc_regul, edge_type):
    """what a helper evaluates for one image (also the in-process fallback of :class:`HostMathPool`)"""
    from pyimsegm_amd import graph_cuts as G
    proba = G.predict_proba(model, features)
    unary = G.compute_unary_cost(proba)
    pairwise = G.compute_pairwise_cost(gc_regul, proba.shape)
    weights = G.edge_weights_from_graph(edges, centres, features, proba, edge_type)
    return proba, unary, pairwise, weights


def _serve():
    """main loop of a helper process"""
    stdin, stdout = sys.stdin.buffer, sys.stdout.buffer
    sys.stdout = sys.stderr                      # stray prints must not corrupt the protocol
    try:
        from threadpoolctl import threadpool_limits
        threadpool_limits(limits=1)
    except Exception:
        pass
    model = None
    while True:
        try:
            msg = _recv(stdin)
        except EOFError:
            return
        try:
            if msg[0] == 'model':
                model = msg[1]
                out = ('ok', None)
            elif msg[0] == 'terms':
                out = ('ok', graph_cut_terms(model, *msg[1:]))
            elif msg[0] == 'quit':
                return
            else:
                out = ('error', 'unknown request %r' % (msg[0], ))
        except Exception as ex:                  # report, keep serving
            out = ('error', '%s: %s' % (type(ex).__name__, ex))
        _send(stdout, out)


class HostMathPool(object):
    """``nb_workers`` helper processes; every call borrows one of them (blocking while all are busy)"""

    def __init__(self, nb_workers):
        self._pid = os.getpid()                   # a forked child inherits the handles but must not talk to our helpers
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        env = dict(os.environ)
        env['PYTHONPATH'] = root + os.pathsep + env.get('PYTHONPATH', '')
        for name in ('OMP_NUM_THREADS', 'OPENBLAS_NUM_THREADS', 'MKL_NUM_THREADS'):
            env[name] = '1'
        self._procs = [subprocess.Popen([sys.executable, '-c', 'from pyimsegm_amd.hostpool import _serve; _serve()'],
                                        stdin=subprocess.PIPE, stdout=subprocess.PIPE, env=env)
                       for _ in range(max(1, int(nb_workers)))]
        for proc in self._procs:                 # a message (~0.25 MB) should fit the pipe: fewer wake-ups per round trip
            for stream in (proc.stdin, proc.stdout):
                try:
                    import fcntl
                    fcntl.fcntl(stream.fileno(), getattr(fcntl, 'F_SETPIPE_SZ', 1031), 1 << 20)
                except Exception:
                    pass
        self._free = queue.Queue()
        for proc in self._procs:
            self._free.put(proc)
        self._lock = threading.Lock()

    def _call(self, proc, msg):
        _send(proc.stdin, msg)
        status, payload = _recv(proc.stdout)
        if status != 'ok':
            raise RuntimeError('host helper failed: %s' % payload)
        return payload

    def set_model(self, model):
        """hand the fitted model to every helper (call while no image is in flight)"""
        with self._lock:
            for proc in self._procs:
                self._call(proc, ('model', model))

    def terms(self, features, edges, centres, gc_regul, edge_type):
        """proba, unary cost, pairwise cost and edge weights of one image, evaluated by a free helper"""
        if self._pid != os.getpid():
            raise RuntimeError('this pool belongs to the parent process: create a pool after the fork')
        proc = self._free.get()
        try:
            return self._call(proc, ('terms', features, edges, centres, gc_regul, edge_type))
        finally:
            self._free.put(proc)

    def close(self):
        if getattr(self, '_pid', None) != os.getpid():
            self._procs = []
            return
        for proc in self._procs:
            try:
                _send(proc.stdin, ('quit', ))
                proc.stdin.close()
            except Exception:
                pass
        for proc in self._procs:
            try:
                proc.wait(timeout=5)
            except Exception:
                proc.kill()
        self._procs = []

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_shared = {}
_shared_lock = threading.Lock()


def shared_pool(nb_workers):
    """process-wide pool of ``nb_workers`` helpers, started at first use (about 2 s: the helpers import scikit-learn)
    and kept for later batches; closed at interpreter exit"""
    key = (os.getpid(), int(nb_workers))
    with _shared_lock:
        pool = _shared.get(key)
        if pool is None or not pool._procs or any(p.poll() is not None for p in pool._procs):
            pool = HostMathPool(nb_workers)
            _shared[key] = pool
        return pool


def _close_shared():
    for pool in list(_shared.values()):
        try:
            pool.close()
        except Exception:
            pass
    _shared.clear()


import atexit  # noqa: E402

atexit.register(_close_shared)
