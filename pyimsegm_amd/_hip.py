"""Thin ctypes binding of ``libimsegm_hip.so`` (C ABI declared in ``include/imsegm_hip.h``).

The library holds every compute kernel of the package (hand-written HIP for gfx950).  There is no
CPU fallback: if the shared library or a GPU is missing, the calls raise ``HipUnavailableError``.
The HIP runtime is initialised lazily, per process, on the first call that needs the device, so
importing the package before ``multiprocessing`` forks workers (the reference's only parallelism,
``imsegm/utilities/experiments.py:392-403``) is safe.
"""
import ctypes as C
import functools
import os
import threading

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('IMSEGM_HIP_LIBRARY') or os.path.join(_HERE, 'libimsegm_hip.so')

U8, F64, F32 = 0, 1, 2
PROFILE_GROUPS = {
    'slic_assign': 0,
    'slic': 1,
    'connectivity': 2,
    'color_stats': 3,
    'graph': 4,
    'graphcut': 5,
    'gather': 6,
    'slic_preprocess': 7,
    'terms': 8,
    'texture': 9,
}


class HipUnavailableError(RuntimeError):
    """the HIP library or a HIP device is not available (no CPU fallback exists)"""


class HipError(RuntimeError):
    """a C-ABI call returned an error status"""


class HipFusedPathError(HipError):
    """``IMSEGM_E_FUSED_PATH`` of include/imsegm_hip.h: the fused back half (``imsegm_image2d_segment`` /
    ``imsegm_image2d_graph_prepare``) does not apply to this label map on this device -- nothing is wrong with the session, the
    caller takes the staged calls (graph, terms, ``cut_general_graph``, gather) instead"""


IMSEGM_E_FUSED_PATH = -3


_lib = None
_lib_lock = threading.Lock()

_vp = C.c_void_p
_ip = C.POINTER(C.c_int)

class GmmParams(C.Structure):
    """``imsegm_gmm`` of include/imsegm_hip.h"""
    _fields_ = [('n_features', C.c_int), ('n_classes', C.c_int), ('scaler_mean', _vp), ('scaler_scale', _vp),
                ('prec_chol', _vp), ('mu_proj', _vp), ('log_det', _vp), ('log_weights', _vp), ('const_term', C.c_double)]


class TermsDebug(C.Structure):
    """``imsegm_terms_debug`` of include/imsegm_hip.h"""
    _fields_ = [('edge_capacity', C.c_int), ('n_edges', C.c_int), ('edges', _vp), ('edge_weights', _vp),
                ('edge_weights_int', _vp), ('unary', _vp), ('unary_int', _vp), ('centres', _vp), ('energy', _vp),
                ('keep_soft_on_device', C.c_int), ('segm_u8', C.c_int), ('soft_f32', C.c_int)]


#: edge types of ``imsegm_image2d_segment``; 0x100 = divide by the relative centre distance (graph_cuts.py:647-650)
EDGE_TYPES = {'': 0, 'const': 0, 'spatial': 1 | 0x100, 'model': 2 | 0x100, 'model_lT': 2, 'model_l1': 3, 'model_l2': 4,
              'features': 5 | 0x100}

_SIGNATURES = {
    'imsegm_last_error': (C.c_char_p, []),
    'imsegm_host_alloc': (C.c_int, [C.c_size_t, C.POINTER(_vp)]),
    'imsegm_host_free': (None, [_vp]),
    'imsegm_device_alloc': (C.c_int, [C.c_int, C.c_size_t, C.POINTER(_vp)]),
    'imsegm_device_free': (None, [_vp]),
    'imsegm_set_device': (C.c_int, [C.c_int]),
    'imsegm_ctx_stream': (C.c_int, [_vp, C.POINTER(_vp)]),
    'imsegm_ctx_copy': (C.c_int, [_vp, _vp, _vp, C.c_size_t, C.c_int]),
    'imsegm_image2d_features_color': (C.c_int, [_vp, C.c_int, _vp]),
    'imsegm_image2d_features_place': (C.c_int, [_vp, C.c_int, C.c_int]),
    'imsegm_image2d_get_features': (C.c_int, [_vp, _vp, C.c_int]),
    'imsegm_image2d_run_color': (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.c_int, C.c_double, _vp, C.c_int, C.c_int, C.c_int, C.c_int,
                                           C.c_int, C.POINTER(GmmParams), C.c_int, _vp, C.c_int, C.c_double, C.c_int, _vp, _vp, _vp, _ip]),
    'imsegm_device_mem_info': (C.c_int, [C.c_int, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
    'imsegm_image2d_graph_prepare': (C.c_int, [_vp]),
    'imsegm_image2d_segment': (C.c_int, [_vp, C.POINTER(GmmParams), _vp, C.c_int, _vp, C.c_int, C.c_double, C.c_int, _vp, _vp, _vp,
                                         _vp, _vp, C.POINTER(TermsDebug)]),
    'imsegm_version': (C.c_int, []),
    'imsegm_device_count': (C.c_int, [_ip]),
    'imsegm_ctx_create': (C.c_int, [C.c_int, C.POINTER(_vp)]),
    'imsegm_ctx_destroy': (None, [_vp]),
    'imsegm_ctx_synchronize': (C.c_int, [_vp]),
    'imsegm_ctx_profile_enable': (C.c_int, [_vp, C.c_int]),
    'imsegm_ctx_profile_reset': (C.c_int, [_vp]),
    'imsegm_ctx_profile_get': (C.c_int, [_vp, C.c_int, C.POINTER(C.c_double), _ip]),
    'imsegm_image2d_create': (C.c_int, [_vp, C.c_int, C.c_int, C.POINTER(_vp)]),
    'imsegm_image2d_destroy': (None, [_vp]),
    'imsegm_image2d_upload': (C.c_int, [_vp, _vp, C.c_int]),
    'imsegm_image2d_slic': (C.c_int, [_vp, C.c_int, C.c_int, C.c_double, _vp, C.c_int, _vp, C.c_int, _vp, C.c_int,
                                      C.c_int, C.c_int, C.c_double, C.c_double, C.c_int, C.c_int, C.c_int, _ip]),
    'imsegm_image2d_get_labels': (C.c_int, [_vp, _vp]),
    'imsegm_image2d_set_labels': (C.c_int, [_vp, _vp, C.c_int]),
    'imsegm_image2d_enforce_connectivity': (C.c_int, [_vp, _vp, C.c_long, C.c_long, C.c_int, _vp]),
    'imsegm_debug_conn_general_runs': (C.c_long, []),
    'imsegm_debug_slic_sweep_runs': (C.c_int, [_vp, _vp]),
    'imsegm_debug_gc_grid_fallbacks': (C.c_long, []),
    'imsegm_assume_bg_on_boundary': (C.c_int, [_vp, _vp, C.c_int, C.c_int, _vp, C.c_int, _ip]),
    'imsegm_image2d_all_finite': (C.c_int, [_vp, _ip]),
    'imsegm_image2d_lm_features': (C.c_int, [_vp, _vp, _vp, C.c_int, C.c_int, C.c_double, C.c_int, _vp]),
    'imsegm_batch2d_create': (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, C.POINTER(_vp)]),
    'imsegm_batch2d_destroy': (None, [_vp]),
    'imsegm_batch2d_run_color': (C.c_int, [_vp, C.c_int, _vp, C.c_int, C.c_int, C.c_int, C.c_double, _vp, C.c_int, C.c_int, C.c_int,
                                           C.c_int, _vp, C.c_int, _vp, C.c_int, C.c_double, C.c_int, _vp, _vp, _vp]),
    'imsegm_batch2d_device_ptr': (C.c_int, [_vp, C.c_int, C.c_int, C.POINTER(_vp)]),
    'imsegm_device_pci_bus_id': (C.c_int, [C.c_int, C.c_char_p, C.c_int]),
    'imsegm_init': (C.c_int, [C.c_int]),
    'imsegm_debug_reload_env': (None, []),
    'imsegm_image2d_label_hist': (C.c_int, [_vp, _vp, C.c_int, _vp]),
    'imsegm_image2d_get_lab': (C.c_int, [_vp, _vp]),
    'imsegm_image2d_get_nearest': (C.c_int, [_vp, _vp]),
    'imsegm_image2d_color_stats': (C.c_int, [_vp, _vp, _vp, _vp]),
    'imsegm_image2d_graph': (C.c_int, [_vp, _vp, C.c_int, _ip, _vp, _vp]),
    'imsegm_image2d_gather': (C.c_int, [_vp, _vp, _vp, C.c_int, _vp, _vp]),
    'imsegm_image2d_lm_prepare': (C.c_int, [_vp, _vp, C.c_int, _vp]),
    'imsegm_image2d_lm_features_sep': (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_int, C.c_int, C.c_double, C.c_int, _vp]),
    'imsegm_image2d_lm_battery': (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.c_double, C.POINTER(C.c_double)]),
    'imsegm_image2d_response_stats': (C.c_int, [_vp, C.c_double, C.c_double, _vp, _vp, _vp]),
    'imsegm_image2d_get_response': (C.c_int, [_vp, _vp]),
    'imsegm_image2d_device_ptr': (C.c_int, [_vp, C.c_int, C.POINTER(_vp)]),
    'imsegm_volume_create': (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, C.POINTER(_vp)]),
    'imsegm_volume_upload': (C.c_int, [_vp, _vp, C.c_int, C.c_double, C.c_double]),
    'imsegm_volume_slic': (C.c_int, [_vp, C.c_int, C.c_double, _vp, C.c_int, _vp, C.c_int, _vp, C.c_int, _vp, C.c_int,
                                     C.c_int, C.c_double, C.c_double, C.c_int, _ip]),
    'imsegm_volume_label_cc': (C.c_int, [_vp, _ip]),
    'imsegm_volume_gray_stats': (C.c_int, [_vp, _vp, _vp, _vp]),
    'imsegm_volume_graph': (C.c_int, [_vp, _vp, C.c_int, _ip, _vp, _vp]),
    'imsegm_image2d_median': (C.c_int, [_vp, _vp]),
    'imsegm_image2d_mean_gradient': (C.c_int, [_vp, _vp]),
    'imsegm_label_hist2d': (C.c_int, [_vp, _vp, C.c_int, C.c_int, _vp, C.c_int, _vp, C.c_int, C.c_int, C.c_int, _vp]),
    'imsegm_ray_features_binary2d': (C.c_int, [_vp, _vp, C.c_int, C.c_int, _vp, C.c_int, _vp, C.c_int, C.c_int, _vp]),
    'imsegm_cut_general_graph': (C.c_int, [_vp, _vp, C.c_int, _vp, _vp, C.c_int, C.c_int, _vp, C.c_int, _vp,
                                           C.POINTER(C.c_int64)]),
}

#: every symbol ``include/imsegm_hip.h`` declares
EXPORTED_SYMBOLS = tuple(_SIGNATURES)


def load_library():
    """load ``libimsegm_hip.so`` and declare the signatures (no device is touched)"""
    global _lib
    with _lib_lock:
        if _lib is None:
            if not os.path.exists(LIB_PATH):
                raise HipUnavailableError(
                    'libimsegm_hip.so is not built (%s); run `python -m pyimsegm_amd.build`' % LIB_PATH)
            try:
                lib = C.CDLL(LIB_PATH)
            except OSError as ex:
                raise HipUnavailableError('cannot load %s: %s' % (LIB_PATH, ex))
            for name, (res, args) in _SIGNATURES.items():
                fn = getattr(lib, name)
                fn.restype = res
                fn.argtypes = args
            _lib = lib
    return _lib


def init(hardware_queues=8):
    """Explicit, optional, once per process before the first device call (``imsegm_init``): the number of hardware queues the HIP
    runtime maps this process's streams onto (its ``GPU_MAX_HW_QUEUES``, default 4).  With one stream per image in flight plus
    the default stream, a fourth image shares a queue with another one and its kernels wait behind that image's: eight queues
    gave 6.5-6.9 instead of 5.6-5.8 Gpx/s on the 2048^2 line with four images in flight (round 3).
    Returns True when the request was recorded; False when the runtime is already up or the user has set the variable (his
    value wins).  Loading the library never touches the environment."""
    status = load_library().imsegm_init(int(hardware_queues))
    if status < 0:
        raise HipError(load_library().imsegm_last_error().decode('utf-8', 'replace'))
    if status == 0:
        # (the C library's setenv does not show in os.environ, which Python copied at start-up: keep the two in step for
        # child processes started with os.environ and for anyone who looks)
        os.environ.setdefault('GPU_MAX_HW_QUEUES', str(int(hardware_queues)))
    return status == 0


def device_pci_bus_id(device=0):
    """PCI address of a HIP device, e.g. ``'0000:c1:00.0'``"""
    buf = C.create_string_buffer(64)
    _check(load_library().imsegm_device_pci_bus_id(int(device), buf, 64))
    return buf.value.decode('ascii', 'replace').lower()


def mem_info(device=0):
    """(free, total) bytes of a device's memory right now"""
    free_b, total_b = C.c_size_t(0), C.c_size_t(0)
    _check(load_library().imsegm_device_mem_info(int(device), C.byref(free_b), C.byref(total_b)))
    return int(free_b.value), int(total_b.value)


def reload_env():
    """read the IMSEGM_* debug switches again (the library reads them once): tests flip them at run time"""
    load_library().imsegm_debug_reload_env()


def _check(status):
    if status != 0:
        message = load_library().imsegm_last_error().decode('utf-8', 'replace')
        raise (HipFusedPathError if status == IMSEGM_E_FUSED_PATH else HipError)(message)


_device_count = {}


def device_count():
    """number of visible HIP devices (cached per process: forked children ask again)"""
    pid = os.getpid()
    if pid not in _device_count:
        n = C.c_int(0)
        load_library().imsegm_device_count(C.byref(n))
        _device_count[pid] = n.value
    return _device_count[pid]


def _ptr(arr):
    return None if arr is None else arr.ctypes.data_as(_vp)


class Context(object):
    """one HIP device + one stream"""

    #: sessions alive on this context (a context with sessions is never released behind their back)
    users = 0

    def __init__(self, device=0):
        #: idle sessions kept for reuse, by image shape (superpixels._open_session / _release_session)
        self.idle_sessions = {}
        lib = load_library()
        if device_count() < 1:
            raise HipUnavailableError('no HIP device is visible: the imsegm HIP path cannot run (no CPU fallback)')
        self._h = _vp()
        self.device = device
        self.pid = os.getpid()
        _check(lib.imsegm_ctx_create(device, C.byref(self._h)))

    def synchronize(self):
        _check(load_library().imsegm_ctx_synchronize(self._h))

    @property
    def stream(self):
        """the HIP stream of this context as an integer handle (for RCCL calls on the same stream)"""
        p = _vp()
        _check(load_library().imsegm_ctx_stream(self._h, C.byref(p)))
        return p.value

    def copy(self, dst_ptr, src_ptr, nbytes, synchronize=True):
        """``hipMemcpyAsync`` (any direction) on this context's stream"""
        _check(load_library().imsegm_ctx_copy(self._h, _vp(dst_ptr), _vp(src_ptr), int(nbytes), int(bool(synchronize))))

    def profile_enable(self, enable=True):
        _check(load_library().imsegm_ctx_profile_enable(self._h, int(enable)))

    def profile_reset(self):
        _check(load_library().imsegm_ctx_profile_reset(self._h))

    def profile_get(self, group):
        ms, n = C.c_double(0), C.c_int(0)
        _check(load_library().imsegm_ctx_profile_get(self._h, PROFILE_GROUPS[group], C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def close_idle_sessions(self):
        """give the device memory of the sessions kept for reuse back (a volume session owns ~70 bytes per voxel)"""
        for sess in list(self.idle_sessions.values()):
            sess.close()
        self.idle_sessions.clear()

    def close(self):
        self.close_idle_sessions()
        if self._h and self.pid == os.getpid():
            load_library().imsegm_ctx_destroy(self._h)
        self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_default_ctx = {}
_default_ctx_lock = threading.RLock()


def _reap_contexts_locked(pid):
    alive = {t.ident for t in threading.enumerate()}
    dead = [k for k in _default_ctx if k[0] == pid and k[1] not in alive
            and _default_ctx[k].users == len(_default_ctx[k].idle_sessions)]
    for old in dead:
        _default_ctx.pop(old).close()
    return len(dead)


def reap_contexts():
    """close the contexts of this process's threads that have ended, with the sessions they kept for reuse (a gray-volume session
    holds ~75 bytes of device memory per voxel): done whenever a new thread asks for its context, and here on request -- after a
    pool of worker threads has gone and before the calling thread needs the memory itself.  Returns the number closed."""
    with _default_ctx_lock:
        return _reap_contexts_locked(os.getpid())


def default_context():
    """per-process, per-thread, per-device lazily created context (re-created in forked children).

    A context owns one HIP stream and its pinned staging buffers, so worker threads that keep several
    images in flight on one GPU (``pipelines.NB_WORKERS``) each get their own."""
    key = (os.getpid(), threading.get_ident(), os.environ.get('IMSEGM_HIP_DEVICE', os.environ.get('LOCAL_RANK', '0')))
    # the whole lookup runs under the lock: thread identifiers are recycled, and a new thread must not pick up the
    # context of a dead thread with the same identifier while another new thread is giving that context back
    with _default_ctx_lock:
        ctx = _default_ctx.get(key)
        if ctx is None:
            # a new thread asks for its context: first give back those of threads that have ended (and the
            # sessions they kept for reuse)
            _reap_contexts_locked(key[0])
            device = int(key[2])
            n = device_count()
            if n > 0:
                device %= n
            ctx = Context(device)
            _default_ctx[key] = ctx
    return ctx


class _PinnedBlock(object):
    """a page-locked host allocation; goes back to the free list of its size class when the last numpy view dies"""
    __slots__ = ('ptr', 'size', 'pid', '__weakref__')

    def __init__(self, ptr, size):
        self.ptr, self.size, self.pid = ptr, size, os.getpid()

    def __del__(self):
        try:
            if self.pid != os.getpid():
                return
            with _pinned_lock:
                cache = _pinned_free.setdefault(self.size, [])
                if len(cache) < (_PINNED_KEEP if self.size < _PINNED_BIG else _PINNED_KEEP_BIG):
                    cache.append(self.ptr)
                    return
            load_library().imsegm_host_free(self.ptr)
        except Exception:
            pass


_pinned_free = {}
_pinned_lock = threading.RLock()
_PINNED_KEEP = 16      # blocks kept per size class for reuse (hipHostMalloc costs far more than the copy it speeds up)
_PINNED_BIG, _PINNED_KEEP_BIG = 256 << 20, 3      # ... of 256 MB and more (the 4.3 GB class map of a 64 x 4096 x 4096 volume): three


def pinned_empty(shape, dtype):
    """``numpy.empty(shape, dtype)`` in page-locked host memory: uploads from / downloads into such an array run as
    asynchronous DMA (``Image2D.upload``, ``Image2D.segment``).  The memory returns to a small per-size free list when
    the array (and every view of it) is garbage collected."""
    dtype = np.dtype(dtype)
    shape = tuple(int(v) for v in np.atleast_1d(shape))
    nbytes = int(np.prod(shape, dtype=np.int64)) * dtype.itemsize
    size = max(4096, (nbytes + 4095) & ~4095)
    ptr = None
    with _pinned_lock:
        cache = _pinned_free.get(size)
        if cache:
            ptr = cache.pop()
    if ptr is None:
        if device_count() < 1:
            raise HipUnavailableError('no HIP device is visible: page-locked memory is not available')
        p = _vp()
        _check(load_library().imsegm_host_alloc(size, C.byref(p)))
        ptr = p.value
    block = _PinnedBlock(ptr, size)
    buf = (C.c_char * size).from_address(ptr)
    buf._block = block                                    # the ctypes buffer (base of the array) keeps the block alive
    return np.frombuffer(buf, dtype=dtype, count=nbytes // dtype.itemsize).reshape(shape)


class DeviceGmm(object):
    """the constants of a fitted ``Pipeline([StandardScaler,] GaussianMixture(covariance_type='full'))`` that the
    device needs for ``predict_proba``: what scikit-learn computes once per model (``means_ @ precisions_cholesky_``,
    the log-determinants, ``log(weights_)``), formed with the very numpy expressions of ``sklearn/mixture/
    _gaussian_mixture.py`` (``_estimate_log_gaussian_prob``, ``_compute_log_det_cholesky``)"""

    def __init__(self, model):
        from sklearn.mixture import GaussianMixture
        from sklearn.pipeline import Pipeline
        from sklearn.preprocessing import StandardScaler
        steps = list(model.steps) if isinstance(model, Pipeline) else [('model', model)]
        gmm = steps[-1][1]
        if type(gmm) is not GaussianMixture or gmm.covariance_type != 'full' or not hasattr(gmm, 'precisions_cholesky_'):
            raise TypeError('not a fitted full-covariance GaussianMixture')
        if len(steps) > 2 or any(type(st) is not StandardScaler for _, st in steps[:-1]):
            raise TypeError('only an optional StandardScaler in front of the mixture is evaluated on the device')
        n_comp, n_feat = gmm.means_.shape
        if n_feat > 256 or n_comp > 16:
            raise TypeError('device class model: at most 256 features and 16 classes')
        self.n_features, self.n_classes = int(n_feat), int(n_comp)
        self.classes = getattr(model, 'classes_', None)
        par = GmmParams()
        par.n_features, par.n_classes = self.n_features, self.n_classes
        self.scaler_mean = self.scaler_scale = None
        if len(steps) == 2:
            scaler = steps[0][1]
            if scaler.with_mean:
                self.scaler_mean = np.ascontiguousarray(scaler.mean_, dtype=np.float64)
            if scaler.with_std:
                self.scaler_scale = np.ascontiguousarray(scaler.scale_, dtype=np.float64)
        chol = np.ascontiguousarray(gmm.precisions_cholesky_, dtype=np.float64)
        self.prec_chol = chol
        self.mu_proj = np.ascontiguousarray([np.dot(mu, pc) for mu, pc in zip(gmm.means_, chol)], dtype=np.float64)
        self.log_det = np.ascontiguousarray(np.sum(np.log(chol.reshape(n_comp, -1)[:, ::n_feat + 1]), 1), dtype=np.float64)
        self.log_weights = np.ascontiguousarray(np.log(gmm.weights_), dtype=np.float64)
        self.const_term = float(n_feat * np.log(2 * np.pi))
        for name in ('scaler_mean', 'scaler_scale', 'prec_chol', 'mu_proj', 'log_det', 'log_weights'):
            setattr(par, name, _ptr(getattr(self, name)))
        par.const_term = self.const_term
        self.params = par


@functools.lru_cache(maxsize=64)
def gaussian_taps(sigma, truncate=4.0):
    """half of the kernel of ``scipy.ndimage.gaussian_filter1d(sigma)`` (taps[0] = centre), computed
    with the very numpy expressions of ``scipy.ndimage._filters._gaussian_kernel1d``; None: no blur
    (cached: the result must be treated as read-only)"""
    sigma = float(sigma)
    if not sigma > 0:
        return None
    radius = int(truncate * sigma + 0.5)
    sigma2 = sigma * sigma
    x = np.arange(-radius, radius + 1)
    phi_x = np.exp(-0.5 / sigma2 * x**2)
    phi_x = phi_x / phi_x.sum()
    return np.ascontiguousarray(phi_x[::-1][radius:], dtype=np.float64)


_DTYPES = {np.dtype(np.uint8): U8, np.dtype(np.float64): F64, np.dtype(np.float32): F32}


class Image2D(object):
    """device-resident pipeline state of one H x W colour image"""

    def __init__(self, height, width, ctx=None):
        self.ctx = ctx or default_context()
        self.shape = (int(height), int(width))
        self._h = _vp()
        self.n_labels = 0
        _check(load_library().imsegm_image2d_create(self.ctx._h, self.shape[0], self.shape[1], C.byref(self._h)))
        self.ctx.users += 1

    def close(self):
        if self._h and self.ctx is not None:
            self.ctx.users -= 1
            if self.ctx._h and self.ctx.pid == os.getpid():
                load_library().imsegm_image2d_destroy(self._h)
        self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def upload(self, image):
        """H x W x 3 image; uint8 / float32 / float64 go up as they are, anything else as float64"""
        image = np.asarray(image)
        if image.ndim != 3 or image.shape[2] != 3 or image.shape[:2] != self.shape:
            raise ValueError('expected an image of shape %r + (3,), got %r' % (self.shape, image.shape))
        if image.dtype not in _DTYPES:
            image = image.astype(np.float64)
        image = np.ascontiguousarray(image)
        _check(load_library().imsegm_image2d_upload(self._h, _ptr(image), _DTYPES[image.dtype]))
        self._uploaded = image          # a page-locked source is read asynchronously: keep it alive until the next sync
        return self

    def run_color(self, image, n_segments, compactness, gmm, pairwise, edge_type='model', feature_flags=(True, True, True),
                  sigma=1., normalize=2, max_iter=10, start_label=0, slic_zero=False, edge_cost=1., use_graphcut=True,
                  classes=None, want_soft=False, pinned=True):
        """upload + SLIC + colour features + class model + graph cut + gathers of one uint8 / float64 colour image in
        one C call (``imsegm_image2d_run_color``); returns (segm int32 H x W, soft or None)"""
        image = np.ascontiguousarray(image)
        if image.shape != self.shape + (3, ) or image.dtype not in _DTYPES:
            raise ValueError('expected an image of shape %r + (3,) and dtype uint8 / float32 / float64' % (self.shape, ))
        code = EDGE_TYPES.get(edge_type)
        if code is None:
            raise ValueError('edge type %r is not evaluated on the device' % (edge_type, ))
        pairwise = np.ascontiguousarray(pairwise, dtype=np.float64)
        nc = gmm.n_classes
        if pairwise.shape != (nc, nc):
            raise ValueError('pairwise cost must be %d x %d' % (nc, nc))
        cl = None if classes is None else np.ascontiguousarray(classes, dtype=np.int32)
        taps = gaussian_taps(sigma)
        r = -1 if taps is None else len(taps) - 1
        alloc = pinned_empty if pinned else np.empty
        segm = alloc(self.shape, np.int32)
        soft = alloc(self.shape + (nc, ), np.float64) if want_soft else None
        mask = (1 if feature_flags[0] else 0) | (2 if feature_flags[1] else 0) | (4 if feature_flags[2] else 0)
        n_out = C.c_int(0)
        _check(load_library().imsegm_image2d_run_color(
            self._h, _ptr(image), _DTYPES[image.dtype], int(normalize), int(n_segments), float(compactness), _ptr(taps), r,
            int(max_iter), int(start_label), int(bool(slic_zero)), mask, C.byref(gmm.params), nc, _ptr(pairwise), code,
            float(edge_cost), int(bool(use_graphcut)), _ptr(cl), _ptr(segm), _ptr(soft), C.byref(n_out)))
        self.n_labels = n_out.value
        self._uploaded = image
        return segm, soft

    def median(self):
        """per-label median of the uploaded image / volume on the current labels: K x 3 (K for a volume)"""
        out = np.empty((self.n_labels, 3) if len(self.shape) == 2 else (self.n_labels, ), dtype=np.float64)
        _check(load_library().imsegm_image2d_median(self._h, _ptr(out)))
        return out

    def mean_gradient(self):
        """per-label mean of ``np.sum(np.gradient(slice), axis=0)`` (stored in the image's dtype): K x 3 (K for a volume)"""
        out = np.empty((self.n_labels, 3) if len(self.shape) == 2 else (self.n_labels, ), dtype=np.float64)
        _check(load_library().imsegm_image2d_mean_gradient(self._h, _ptr(out)))
        return out

    def features_color(self, mean=True, std=True, energy=True, to_host=True):
        """resident feature table (columns mean | std | energy, 3 each) of the uploaded image on the current labels;
        returns it as K x F float64 when ``to_host``"""
        mask = (1 if mean else 0) | (2 if std else 0) | (4 if energy else 0)
        nflags = bool(mean) + bool(std) + bool(energy)
        out = np.empty((self.n_labels, 3 * nflags), dtype=np.float64) if to_host else None
        _check(load_library().imsegm_image2d_features_color(self._h, mask, _ptr(out)))
        return out

    def graph_prepare(self):
        """enqueue the graph of the resident label map (neighbour pairs, centres, edges, arcs) ahead of :meth:`segment`, without
        a synchronisation (``imsegm_image2d_graph_prepare``): it depends on the label map alone, so it can run while the host fits
        the class model.  Raises :class:`HipFusedPathError` when the fused back half does not apply."""
        _check(load_library().imsegm_image2d_graph_prepare(self._h))

    def segment(self, pairwise, edge_type='model', edge_cost=1., gmm=None, proba=None, use_graphcut=True, classes=None,
                want_segm=True, want_soft=False, want_graph_labels=False, want_proba=False, debug=False, pinned=True,
                keep_soft_on_device=False, segm_dtype=None, soft_dtype=None, segm_out=None):
        """fused back half of the pipeline on the resident label map (``imsegm_image2d_segment``): class probabilities
        (``gmm``: :class:`DeviceGmm` on the resident features, else ``proba`` K x C from the host), unary / edge terms,
        alpha-expansion, ``classes[graph_labels][slic]`` and ``proba[slic]``; one synchronisation.

        ``segm_out``: the caller's own C-contiguous array for the class map (shape and dtype of 'segm'), e.g. one whose pages have been
        touched while the device worked -- a download into untouched pageable memory runs at half the rate of the link.

        :return dict: 'segm' (H x W int32), 'soft' (H x W x C), 'graph_labels' (K), 'proba' (K x C) as requested, plus
            with ``debug`` the graph-cut terms ('edges', 'edge_weights', 'edge_weights_int', 'unary', 'unary_int',
            'centres', 'energy')"""
        code = EDGE_TYPES.get(edge_type)
        if code is None:
            raise ValueError('edge type %r is not evaluated on the device' % (edge_type, ))
        pairwise = np.ascontiguousarray(pairwise, dtype=np.float64)
        nc = pairwise.shape[0]
        if pairwise.shape != (nc, nc):
            raise ValueError('pairwise cost must be square')
        k = self.n_labels
        pr = None
        if gmm is None:
            pr = np.ascontiguousarray(proba, dtype=np.float64)
            if pr.ndim != 2 or pr.shape[0] < k or pr.shape[1] != nc:
                raise ValueError('proba %r does not fit %d superpixels x %d classes' % (pr.shape, k, nc))
            pr = np.ascontiguousarray(pr[:k])
        elif gmm.n_classes != nc:
            raise ValueError('class model has %d classes, pairwise cost %d' % (gmm.n_classes, nc))
        cl = None if classes is None else np.ascontiguousarray(classes, dtype=np.int32)
        if cl is not None and cl.shape != (nc, ):
            raise ValueError('classes must hold one value per class')
        alloc = pinned_empty if pinned else np.empty
        out = {}
        # narrow result formats (explicit opt-in, not the reference's dtypes): uint8 class map, float32 soft segmentation
        segm_u8 = segm_dtype is not None and np.dtype(segm_dtype) == np.uint8
        soft_f32 = soft_dtype is not None and np.dtype(soft_dtype) == np.float32
        if segm_dtype is not None and not segm_u8 and np.dtype(segm_dtype) != np.int32:
            raise ValueError('the class map leaves the device as int32 (the reference) or uint8')
        if soft_dtype is not None and not soft_f32 and np.dtype(soft_dtype) != np.float64:
            raise ValueError('the soft segmentation leaves the device as float64 (the reference) or float32')
        if segm_u8 and (nc > 256 or (cl is not None and (cl.min() < 0 or cl.max() > 255))):
            raise ValueError('uint8 class map: class values must lie in 0..255')
        if want_segm:
            want_dtype = np.dtype(np.uint8 if segm_u8 else np.int32)
            if segm_out is not None:
                if not isinstance(segm_out, np.ndarray) or segm_out.shape != tuple(self.shape) or segm_out.dtype != want_dtype \
                        or not segm_out.flags.c_contiguous or not segm_out.flags.writeable:
                    raise ValueError('segm_out must be a writeable C-contiguous %s array of shape %r' % (want_dtype, tuple(self.shape)))
                out['segm'] = segm_out
            else:
                out['segm'] = alloc(self.shape, want_dtype)
        if want_soft:
            out['soft'] = alloc(self.shape + (nc, ), np.float32 if soft_f32 else np.float64)
        if want_graph_labels or debug:
            out['graph_labels'] = np.empty(k, dtype=np.int32)
        if want_proba or debug:
            out['proba'] = np.empty((k, nc), dtype=np.float64)
        dbg = None
        if debug or keep_soft_on_device or segm_u8 or soft_f32:
            dbg = TermsDebug()
            dbg.keep_soft_on_device = int(bool(keep_soft_on_device))
            dbg.segm_u8, dbg.soft_f32 = int(segm_u8), int(soft_f32)
        if debug:
            cap = (16 if len(self.shape) == 3 else 3) * k + 64
            ndim = len(self.shape)
            out.update(edges=np.empty((cap, 2), np.int32), edge_weights=np.empty(cap), edge_weights_int=np.empty(cap, np.int32),
                       unary=np.empty((k, nc)), unary_int=np.empty((k, nc), np.int32), centres=np.empty((k, ndim)),
                       energy=np.zeros(1, np.int64))
            dbg.edge_capacity = cap
            for name in ('edges', 'edge_weights', 'edge_weights_int', 'unary', 'unary_int', 'centres', 'energy'):
                setattr(dbg, name, _ptr(out[name]))
        _check(load_library().imsegm_image2d_segment(
            self._h, C.byref(gmm.params) if gmm is not None else None, _ptr(pr), nc, _ptr(pairwise), code, float(edge_cost),
            int(bool(use_graphcut)), _ptr(cl), _ptr(out.get('segm')), _ptr(out.get('soft')), _ptr(out.get('graph_labels')),
            _ptr(out.get('proba')), C.byref(dbg) if dbg is not None else None))
        if debug:
            ne = dbg.n_edges
            for name in ('edges', 'edge_weights', 'edge_weights_int'):
                out[name] = out[name][:ne]
            out['energy'] = int(out['energy'][0])
        return out

    def slic(self, n_segments, compactness, sigma=1., normalize=2, max_iter=10, enforce_connectivity=True,
             min_size_factor=0.5, max_size_factor=3., start_label=0, max_candidates=0, slic_zero=False):
        taps = gaussian_taps(sigma)
        r = -1 if taps is None else len(taps) - 1
        n_out = C.c_int(0)
        _check(load_library().imsegm_image2d_slic(
            self._h, int(normalize), int(n_segments), float(compactness), _ptr(taps), r, _ptr(taps), r, _ptr(taps), r,
            int(max_iter), int(bool(enforce_connectivity)), float(min_size_factor), float(max_size_factor),
            int(start_label), int(max_candidates), int(bool(slic_zero)), C.byref(n_out)))
        self.n_labels = n_out.value
        return self.n_labels

    def get_labels(self):
        out = np.empty(self.shape, dtype=np.int64)
        _check(load_library().imsegm_image2d_get_labels(self._h, _ptr(out)))
        return out

    def get_labels_int32(self):
        """the resident label map as it lives in HBM (int32), without the int64 widening of :meth:`get_labels`"""
        out = np.empty(self.shape, dtype=np.int32)
        src = _device_array(self, 0, self.shape, '<i4').__cuda_array_interface__['data'][0]
        self.ctx.copy(out.ctypes.data, src, out.nbytes, synchronize=True)
        return out

    def enforce_connectivity(self, labels, min_size, max_size, start_label=0):
        """``_enforce_label_connectivity_cython`` of scikit-image 0.18 on a given label map; returns the int64 map"""
        labels = np.ascontiguousarray(labels, dtype=np.int32)
        if labels.shape != self.shape:
            raise ValueError('label map %r does not match image %r' % (labels.shape, self.shape))
        n_out = C.c_int(0)
        _check(load_library().imsegm_image2d_enforce_connectivity(self._h, _ptr(labels), int(min_size), int(max_size),
                                                                  int(start_label), C.byref(n_out)))
        self.n_labels = n_out.value
        return self.get_labels()

    def set_labels(self, labels, n_labels=None):
        labels = np.ascontiguousarray(labels, dtype=np.int32)
        if labels.shape != self.shape:
            raise ValueError('label map %r does not match image %r' % (labels.shape, self.shape))
        if labels.size and labels.min() < 0:
            raise ValueError('labels must be non-negative')
        if n_labels is None:
            n_labels = int(labels.max()) + 1
        _check(load_library().imsegm_image2d_set_labels(self._h, _ptr(labels), int(n_labels)))
        self.n_labels = int(n_labels)
        return self

    def label_hist(self, annot, nb_annot=None):
        """counts[k, a] = pixels with resident label k and annotation a (int64 [n_labels, nb_annot])"""
        annot = np.ascontiguousarray(annot, dtype=np.int32)
        if annot.shape != self.shape:
            raise ValueError('annotation %r does not match the session %r' % (annot.shape, self.shape))
        if nb_annot is None:
            nb_annot = int(annot.max()) + 1 if annot.size else 1
        out = np.empty((self.n_labels, int(nb_annot)), dtype=np.int64)
        _check(load_library().imsegm_image2d_label_hist(self._h, _ptr(annot), int(nb_annot), _ptr(out)))
        return out

    def get_lab(self):
        out = np.empty((3,) + self.shape, dtype=np.float64)
        _check(load_library().imsegm_image2d_get_lab(self._h, _ptr(out)))
        return out

    def get_nearest(self):
        out = np.empty(self.shape, dtype=np.int32)
        _check(load_library().imsegm_image2d_get_nearest(self._h, _ptr(out)))
        return out

    def color_stats(self, mean=True, energy=True, var=True):
        k = self.n_labels
        m = np.empty((k, 3), dtype=np.float64) if mean else None
        e = np.empty((k, 3), dtype=np.float64) if energy else None
        v = np.empty((k, 3), dtype=np.float64) if var else None
        _check(load_library().imsegm_image2d_color_stats(self._h, _ptr(m), _ptr(e), _ptr(v)))
        return m, e, v

    # -- Leung-Malik texture responses ------------------------------------------------------------
    def lm_prepare(self, sigma=150.):
        """planes = image - gaussian_filter(image.astype(float), sigma)   (descriptors.py:1078)"""
        taps = gaussian_taps(sigma)
        radius = len(taps) - 1
        full = np.concatenate([taps[:0:-1], taps])
        mix = np.zeros((3, 3))
        for c_out in range(3):          # the same filter along the 3-element channel axis, 'reflect'
            for j in range(-radius, radius + 1):
                i = (c_out + j) % 6
                mix[c_out, i if i < 3 else 5 - i] += full[j + radius]
        _check(load_library().imsegm_image2d_lm_prepare(self._h, _ptr(taps), radius, _ptr(np.ascontiguousarray(mix))))
        return self

    def lm_battery(self, battery, clip):
        """response of one filter battery (k x S x S convolution kernels) -> L2 norm over the channels"""
        battery = np.asarray(battery, dtype=np.float64)
        nk, side = battery.shape[0], battery.shape[1]
        if battery.ndim != 3 or battery.shape[2] != side or side % 2 != 1:
            raise ValueError('wrong battery dim %r' % (battery.shape, ))
        pad = {1: 1, 2: 2, 3: 4, 4: 4, 5: 8, 6: 8, 7: 8, 8: 8}.get(nk)
        if pad is None:
            raise ValueError('at most 8 kernels per battery')
        if pad != nk:                    # repeat the last kernel: the maximum is unchanged
            battery = np.concatenate([battery, np.repeat(battery[-1:], pad - nk, axis=0)], axis=0)
        # true convolution == correlation with the flipped kernel; layout [kx][ky][kernel]
        weights = np.ascontiguousarray(battery[:, ::-1, ::-1].transpose(2, 1, 0))
        ssq = C.c_double(0)
        _check(load_library().imsegm_image2d_lm_battery(self._h, _ptr(weights), pad, side // 2, float(clip), C.byref(ssq)))
        return float(np.sqrt(ssq.value))

    @staticmethod
    def _battery_weights(battery):
        """(weights [kx][ky][kernel] of the flipped kernels, kernels after padding to 1 / 2 / 4 / 8, radius)"""
        battery = np.asarray(battery, dtype=np.float64)
        nk, side = battery.shape[0], battery.shape[1]
        if battery.ndim != 3 or battery.shape[2] != side or side % 2 != 1:
            raise ValueError('wrong battery dim %r' % (battery.shape, ))
        pad = {1: 1, 2: 2, 3: 4, 4: 4, 5: 8, 6: 8, 7: 8, 8: 8}.get(nk)
        if pad is None:
            raise ValueError('at most 8 kernels per battery')
        if pad != nk:                    # repeat the last kernel: the maximum is unchanged
            battery = np.concatenate([battery, np.repeat(battery[-1:], pad - nk, axis=0)], axis=0)
        # true convolution == correlation with the flipped kernel; layout [kx][ky][kernel]
        return np.ascontiguousarray(battery[:, ::-1, ::-1].transpose(2, 1, 0)), pad, side // 2

    #: kernels whose singular values fall below this share of the largest after RANK components are evaluated as separable passes
    SEPARABLE_TOLERANCE, SEPARABLE_MAX_RANK = 1e-13, 2

    #: two dense kernels count as mirror images of each other when they differ by less than this share of their maximum
    MIRROR_TOLERANCE = 1e-12

    @classmethod
    def _mirror_pairs(cls, dense):
        """the dense (flipped) kernels as pairs (a, b, m) with ``dense[b] == m * dense[a][:, ::-1]`` (m = +-1) to MIRROR_TOLERANCE,
        or None when they do not all pair up -- the orientations theta and pi - theta of a Leung-Malik battery do"""
        left = list(range(len(dense)))
        pairs = []
        while left:
            a = left.pop(0)
            mirrored = dense[a][:, ::-1]
            bound = cls.MIRROR_TOLERANCE * np.abs(dense[a]).max()
            for b in left:
                hit = [m for m in (1., -1.) if np.abs(dense[b] - m * mirrored).max() <= bound]
                if hit:
                    pairs.append((a, b, hit[0]))
                    left.remove(b)
                    break
            else:
                return None
        return pairs

    @classmethod
    def _quad_table(cls, dense, pairs, radius):
        """the table of ``k_conv_battery_quad`` (csrc/texture.hip): [x = 0..r][t = 0..r][WS of the pairs | WD of the pairs], WS / WD =
        half the sum / difference of the first kernel of a pair at (row t, column r + x) and (row t, column r - x), the row of
        the kernel centre halved once more; then the mirror signs"""
        r = radius
        table = np.zeros((r + 1, r + 1, 2 * len(pairs)))
        for k, (a, _, _) in enumerate(pairs):
            right, left = dense[a][:r + 1, r:], dense[a][:r + 1, r::-1]           # [t][x]: columns r + x | r - x
            table[:, :, k] = ((right + left) / 2).T
            table[:, :, len(pairs) + k] = ((right - left) / 2).T
        table[:, r, :] /= 2
        return np.concatenate([table.ravel(), [m for _, _, m in pairs]])

    @classmethod
    def _split_battery(cls, battery, separable=True, symmetric=True, mirror=True):
        """one battery (k x S x S convolution kernels) as the device takes it: (dense weights [kx][ky][kernel] of the flipped
        kernels that stay dense, their number after padding to 0 / 1 / 2 / 4 / 6 / 8, separable taps, groups, rank, radius, parity).
        A kernel of numerical rank <= SEPARABLE_MAX_RANK (numpy SVD of the flipped kernel) becomes `rank` pairs of (x taps, y
        taps); at most two kernels per battery go that way (the 0 and 90 degree orientations of an edge / bar battery).
        parity: +-1 the dense kernels are all even / odd under the point reflection; +-2 (``mirror``, side 33) they are mirror
        images of each other in pairs as well -- the dense weights are then the quad table (:meth:`_quad_table`) in a block of
        the usual size."""
        battery = np.asarray(battery, dtype=np.float64)
        nk, side = battery.shape[0], battery.shape[1]
        if battery.ndim != 3 or battery.shape[2] != side or side % 2 != 1:
            raise ValueError('wrong battery dim %r' % (battery.shape, ))
        if nk > 8:
            raise ValueError('at most 8 kernels per battery')
        flipped = battery[:, ::-1, ::-1]
        dense, factors = [], []
        for kernel in flipped:
            parts = None
            if separable and len(factors) < 2:
                u, sv, vt = np.linalg.svd(kernel)
                rank = int(np.sum(sv > cls.SEPARABLE_TOLERANCE * sv[0])) if sv[0] > 0 else 0
                if 1 <= rank <= cls.SEPARABLE_MAX_RANK:
                    parts = [(vt[i], sv[i] * u[:, i]) for i in range(rank)]          # (taps along x, taps along y)
            if parts is None:
                dense.append(kernel)
            else:
                factors.append(parts)
        rank = max([len(p) for p in factors] or [0])
        taps = np.zeros((len(factors), rank, 2, side))
        for g, parts in enumerate(factors):
            for i, (tx, ty) in enumerate(parts):
                taps[g, i, 0], taps[g, i, 1] = tx, ty
        # point symmetry of the kernels that stay dense: all even (K[-p] == K[p]) -> +1, all odd -> -1, bit for bit; else 0
        parity = 0
        if dense and symmetric:
            if all(np.array_equal(k[::-1, ::-1], k) for k in dense):
                parity = 1
            elif all(np.array_equal(k[::-1, ::-1], -k) for k in dense):
                parity = -1
        if parity and mirror and side == 33 and len(dense) in (2, 4, 6, 8):
            pairs = cls._mirror_pairs(dense)
            if pairs is not None:
                weights = np.zeros(side * side * len(dense))
                table = cls._quad_table(dense, pairs, side // 2)
                weights[:table.size] = table
                return weights, len(dense), taps, len(factors), rank, side // 2, 2 * parity
        pad = {0: 0, 1: 1, 2: 2, 3: 4, 4: 4, 5: 6, 6: 6, 7: 8, 8: 8}[len(dense)]
        if pad != len(dense):            # repeat the last kernel: the maximum is unchanged
            dense = dense + [dense[-1]] * (pad - len(dense))
        weights = np.ascontiguousarray(np.asarray(dense).transpose(2, 1, 0)) if dense else np.zeros(0)
        return weights, pad, taps, len(factors), rank, side // 2, parity

    #: the last banks handed to :meth:`lm_features`, split and packed (the factorisation is 60 SVDs per bank)
    _packed_banks = []

    @classmethod
    def _pack_bank(cls, batteries, separable, mirror=True):
        """the arguments of ``imsegm_image2d_lm_features_sep`` for a list of batteries; remembered per list of battery ARRAYS (the
        bank of the descriptors is built once per process) with their sums as a guard against arrays changed in place"""
        batteries = [np.asarray(b, dtype=np.float64) for b in batteries]
        sums = [float(b.sum()) for b in batteries]
        flag_now = (separable, mirror)
        for held, flag, held_sums, packed in cls._packed_banks:
            if flag == flag_now and len(held) == len(batteries) and all(a is b for a, b in zip(held, batteries)) and held_sums == sums:
                return packed
        parts = [cls._split_battery(b, separable, separable, mirror and separable) for b in batteries]
        radius = parts[0][5]
        if any(p[5] != radius for p in parts):
            raise ValueError('the batteries of one call have one kernel size')
        weights = np.concatenate([p[0].ravel() for p in parts])
        taps = np.concatenate([p[2].ravel() for p in parts])
        packed = dict(weights=weights if weights.size else None, taps=taps if taps.size else None, radius=radius, count=len(parts),
                      kernels=np.array([p[1] for p in parts], dtype=np.int32), groups=np.array([p[3] for p in parts], dtype=np.int32),
                      ranks=np.array([p[4] for p in parts], dtype=np.int32), parity=np.array([p[6] for p in parts], dtype=np.int32))
        cls._packed_banks.insert(0, (batteries, flag_now, sums, packed))
        del cls._packed_banks[4:]
        return packed

    def lm_features(self, batteries, clip, mean=True, std=True, energy=True, separable=True, to_host=True, mirror=True):
        """``imsegm_image2d_lm_features_sep``: K x (3 * flags * len(batteries)) statistics of all batteries in one call;
        ``separable=False``: every kernel as a dense S x S sum (``imsegm_image2d_lm_features``); ``to_host=False``: the table
        stays on the device (for :meth:`segment` with a device class model, :meth:`get_features`) and nothing is waited for;
        ``mirror=False``: the dense kernels of a battery one by one (point symmetry only), not as mirror pairs"""
        bank = self._pack_bank(batteries, bool(separable), bool(mirror))
        mask = (1 if mean else 0) | (2 if std else 0) | (4 if energy else 0)
        out = np.empty((self.n_labels, 3 * bin(mask).count('1') * bank['count']), dtype=np.float64) if to_host else None
        _check(load_library().imsegm_image2d_lm_features_sep(
            self._h, _ptr(bank['weights']), _ptr(bank['kernels']), _ptr(bank['parity']), _ptr(bank['taps']), _ptr(bank['groups']),
            _ptr(bank['ranks']), bank['count'], bank['radius'], float(clip), mask, _ptr(out)))
        return out

    def features_place(self, total_columns, column):
        """``imsegm_image2d_features_place``: the next :meth:`features_color` / :meth:`lm_features` (``to_host=False``) writes its
        columns at ``column`` of a resident table ``total_columns`` wide"""
        _check(load_library().imsegm_image2d_features_place(self._h, int(total_columns), int(column)))
        return self

    def get_features(self, columns):
        """the resident K x ``columns`` feature table (``imsegm_image2d_get_features``)"""
        out = np.empty((self.n_labels, int(columns)), dtype=np.float64)
        _check(load_library().imsegm_image2d_get_features(self._h, _ptr(out), int(columns)))
        return out

    def response_stats(self, mul, div, mean=True, energy=True, var=True):
        k = self.n_labels
        m = np.empty((k, 3), dtype=np.float64) if mean else None
        e = np.empty((k, 3), dtype=np.float64) if energy else None
        v = np.empty((k, 3), dtype=np.float64) if var else None
        _check(load_library().imsegm_image2d_response_stats(self._h, float(mul), float(div), _ptr(m), _ptr(e), _ptr(v)))
        return m, e, v

    def get_response(self):
        out = np.empty((3, ) + self.shape, dtype=np.float64)
        _check(load_library().imsegm_image2d_get_response(self._h, _ptr(out)))
        return out

    def graph(self):
        """(edges int32 E x 2 ordered by (b, a); centres K x 2; present flags K)"""
        k = self.n_labels
        cap = max(64, 4 * k)
        centres = np.empty((k, 2), dtype=np.float64)
        present = np.empty(k, dtype=np.uint8)
        while True:
            edges = np.empty((cap, 2), dtype=np.int32)
            ne = C.c_int(0)
            _check(load_library().imsegm_image2d_graph(self._h, _ptr(edges), cap, C.byref(ne), _ptr(centres),
                                                       _ptr(present)))
            if ne.value <= cap:
                return edges[:ne.value], centres, present.astype(bool)
            cap = ne.value

    def all_finite(self):
        """no NaN / inf among the uploaded pixels (``imsegm_image2d_all_finite``)"""
        ok = C.c_int(0)
        _check(load_library().imsegm_image2d_all_finite(self._h, C.byref(ok)))
        return bool(ok.value)

    def gather(self, graph_labels=None, proba=None, to_host=True, segm_out=None):
        """``graph_labels[slic]`` (int32 H x W) and ``proba[slic]`` (float64 H x W x C); ``segm_out``: the caller's own
        C-contiguous int32 array of the session's shape for the first"""
        segm = soft = None
        gl = pr = None
        nc = 0
        if graph_labels is not None:
            gl = np.ascontiguousarray(graph_labels, dtype=np.int32)
            if gl.shape[0] < self.n_labels:
                raise ValueError('label LUT shorter than the number of superpixels')
            segm = np.empty(self.shape, dtype=np.int32) if to_host else None
            if to_host and segm_out is not None:
                if not isinstance(segm_out, np.ndarray) or segm_out.shape != tuple(self.shape) or segm_out.dtype != np.int32 \
                        or not segm_out.flags.c_contiguous or not segm_out.flags.writeable:
                    raise ValueError('segm_out must be a writeable C-contiguous int32 array of shape %r' % (tuple(self.shape), ))
                segm = segm_out
        if proba is not None:
            pr = np.ascontiguousarray(proba, dtype=np.float64)
            if pr.ndim != 2 or pr.shape[0] < self.n_labels:
                raise ValueError('proba LUT shorter than the number of superpixels')
            nc = pr.shape[1]
            soft = np.empty(self.shape + (nc,), dtype=np.float64) if to_host else None
        _check(load_library().imsegm_image2d_gather(self._h, _ptr(gl), _ptr(pr), nc, _ptr(segm), _ptr(soft)))
        return segm, soft


def img_as_float_map(dtype):
    """``skimage.util.img_as_float`` (util/dtype.py) as an affine map ``(v + offset) * scale``"""
    dtype = np.dtype(dtype)
    if dtype.kind in 'fb':
        return 0., 1.
    if dtype.kind == 'u':
        return 0., 1. / np.iinfo(dtype).max
    if dtype.kind == 'i':
        info = np.iinfo(dtype)
        return 0.5, 2. / (float(info.max) - float(info.min))
    raise ValueError('unsupported dtype %r' % dtype)


class Volume3D(Image2D):
    """device-resident state of one D x H x W gray volume (supervoxel path)"""

    def __init__(self, depth, height, width, ctx=None):
        self.ctx = ctx or default_context()
        self.shape = (int(depth), int(height), int(width))
        self._h = _vp()
        self.n_labels = 0
        _check(load_library().imsegm_volume_create(self.ctx._h, self.shape[0], self.shape[1], self.shape[2],
                                                   C.byref(self._h)))
        self.ctx.users += 1

    def upload(self, volume):
        """D x H x W volume; uint8 / float32 / float64 go up as they are, anything else as float64
        (exact for integers up to 53 bits); SLIC sees ``img_as_float`` of the source dtype"""
        volume = np.asarray(volume)
        if volume.shape != self.shape:
            raise ValueError('expected a volume of shape %r, got %r' % (self.shape, volume.shape))
        off, scale = img_as_float_map(volume.dtype)
        if volume.dtype not in _DTYPES:
            volume = volume.astype(np.float64)
        volume = np.ascontiguousarray(volume)
        _check(load_library().imsegm_volume_upload(self._h, _ptr(volume), _DTYPES[volume.dtype], off, scale))
        self.dtype = volume.dtype
        return self

    def slic(self, n_segments, compactness, sigma=1., spacing=(1., 1., 1.), max_iter=10, enforce_connectivity=True,
             min_size_factor=0.5, max_size_factor=3., start_label=0):
        # scikit-image 0.18 keeps spacing and sigma in the dtype of the image: float32 for a float32 volume (which then
        # runs in float32 on the device too), float64 for everything else
        fdt = np.float32 if getattr(self, 'dtype', None) == np.float32 else np.float64
        spacing = np.ascontiguousarray(spacing, dtype=fdt)
        if spacing.shape != (3, ):
            raise ValueError('spacing must have 3 elements (z, y, x)')
        taps = [gaussian_taps(float(s)) for s in np.array([sigma, sigma, sigma], dtype=fdt) / spacing]
        spacing = np.ascontiguousarray(spacing, dtype=np.float64)
        args = []
        for t in taps:
            args += [_ptr(t), -1 if t is None else len(t) - 1]
        n_out = C.c_int(0)
        _check(load_library().imsegm_volume_slic(
            self._h, int(n_segments), float(compactness), *args, _ptr(spacing), int(max_iter),
            int(bool(enforce_connectivity)), float(min_size_factor), float(max_size_factor), int(start_label),
            C.byref(n_out)))
        self.n_labels = n_out.value
        return self.n_labels

    def label_cc(self):
        """``skimage.measure.label`` of the current label map, in place; returns max label + 1"""
        n_out = C.c_int(0)
        _check(load_library().imsegm_volume_label_cc(self._h, C.byref(n_out)))
        self.n_labels = n_out.value
        return self.n_labels

    def gray_stats(self, mean=True, energy=True, var=True):
        k = self.n_labels
        m = np.empty(k, dtype=np.float64) if mean else None
        e = np.empty(k, dtype=np.float64) if energy else None
        v = np.empty(k, dtype=np.float64) if var else None
        _check(load_library().imsegm_volume_gray_stats(self._h, _ptr(m), _ptr(e), _ptr(v)))
        return m, e, v

    def graph(self):
        """(edges int32 E x 2 ordered by (b, a); centres K x 3 (z, y, x); present flags K)"""
        k = self.n_labels
        cap = max(64, 8 * k)
        centres = np.empty((k, 3), dtype=np.float64)
        present = np.empty(k, dtype=np.uint8)
        while True:
            edges = np.empty((cap, 2), dtype=np.int32)
            ne = C.c_int(0)
            _check(load_library().imsegm_volume_graph(self._h, _ptr(edges), cap, C.byref(ne), _ptr(centres),
                                                      _ptr(present)))
            if ne.value <= cap:
                return edges[:ne.value], centres, present.astype(bool)
            cap = ne.value

    def _unsupported(self, *args, **kwargs):
        raise HipError('not available for volumes')

    get_lab = color_stats = run_color = _unsupported

    # -- Leung-Malik responses of the slices (descriptors.py:969-1038: every slice is filtered as a 2-D image) --------
    def lm_prepare(self, sigma=150.):
        """planes = volume - gaussian_filter(slice, sigma) per slice   (image_subtract_gauss_smooth, descriptors.py:981-994)"""
        taps = gaussian_taps(sigma)
        _check(load_library().imsegm_image2d_lm_prepare(self._h, _ptr(taps), len(taps) - 1, None))
        return self

    def response_stats(self, mul, div, mean=True, energy=True, var=True):
        k = self.n_labels
        m = np.empty(k, dtype=np.float64) if mean else None
        e = np.empty(k, dtype=np.float64) if energy else None
        v = np.empty(k, dtype=np.float64) if var else None
        _check(load_library().imsegm_image2d_response_stats(self._h, float(mul), float(div), _ptr(m), _ptr(e), _ptr(v)))
        return m, e, v

    def get_response(self):
        out = np.empty(self.shape, dtype=np.float64)
        _check(load_library().imsegm_image2d_get_response(self._h, _ptr(out)))
        return out


class Batch2D(object):
    """up to ``max_images`` colour images of ONE size through the whole pipeline in one chain of launches
    (``imsegm_batch2d_*``, csrc/batch.hip: image = blockIdx.z): what the reference does by mapping ``segment_image_model``
    over a process pool (``run_segm_slic_model_graphcut.py:505-514``)"""

    def __init__(self, max_images, height, width, ctx=None):
        self.ctx = ctx or default_context()
        self.shape = (int(height), int(width))
        self.max_images = int(max_images)
        self._h = _vp()
        self.n_labels = []
        self.n_images = 0
        _check(load_library().imsegm_batch2d_create(self.ctx._h, self.max_images, self.shape[0], self.shape[1], C.byref(self._h)))
        self.ctx.users += 1

    def close(self):
        if self._h and self.ctx is not None:
            self.ctx.users -= 1
            if self.ctx._h and self.ctx.pid == os.getpid():
                load_library().imsegm_batch2d_destroy(self._h)
        self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def run_color(self, images, n_segments, compactness, gmm, pairwise, edge_type='model', feature_flags=(True, True, True),
                  sigma=1., normalize=2, max_iter=10, start_label=0, edge_cost=1., use_graphcut=True, classes=None, to_host=True,
                  pinned=True, out=None):
        """``Image2D.run_color`` for a list of images of this batch's size and one dtype (uint8 / float32 / float64):
        returns the list of segmentations (int32 H x W; ``out``: arrays to write them into) or, with ``to_host=False``,
        leaves them on the device (:meth:`segm_device_array`)"""
        images = [np.ascontiguousarray(im) for im in images]
        if not 0 < len(images) <= self.max_images:
            raise ValueError('between 1 and %d images per batch' % self.max_images)
        dtype = images[0].dtype
        if dtype not in _DTYPES or any(im.shape != self.shape + (3, ) or im.dtype != dtype for im in images):
            raise ValueError('expected images of shape %r + (3,) and one dtype of uint8 / float32 / float64' % (self.shape, ))
        code = EDGE_TYPES.get(edge_type)
        if code is None:
            raise ValueError('edge type %r is not evaluated on the device' % (edge_type, ))
        pairwise = np.ascontiguousarray(pairwise, dtype=np.float64)
        nc = gmm.n_classes
        if pairwise.shape != (nc, nc):
            raise ValueError('pairwise cost must be %d x %d' % (nc, nc))
        cl = None if classes is None else np.ascontiguousarray(classes, dtype=np.int32)
        taps = gaussian_taps(sigma)
        r = -1 if taps is None else len(taps) - 1
        n = len(images)
        src = (_vp * n)(*[im.ctypes.data for im in images])
        segm, dst = None, None
        if to_host:
            if out is not None:
                segm = list(out)
                if len(segm) != n or any(a.shape != self.shape or a.dtype != np.int32 or not a.flags['C_CONTIGUOUS'] for a in segm):
                    raise ValueError('out: one C-contiguous int32 array of the image size per image')
            else:
                alloc = pinned_empty if pinned else np.empty
                segm = [alloc(self.shape, np.int32) for _ in range(n)]
            dst = (_vp * n)(*[a.ctypes.data for a in segm])
        mask = (1 if feature_flags[0] else 0) | (2 if feature_flags[1] else 0) | (4 if feature_flags[2] else 0)
        counts = (C.c_int * n)()
        _check(load_library().imsegm_batch2d_run_color(
            self._h, n, src, _DTYPES[dtype], int(normalize), int(n_segments), float(compactness), _ptr(taps), r, int(max_iter),
            int(start_label), mask, C.byref(gmm.params), nc, _ptr(pairwise), code, float(edge_cost), int(bool(use_graphcut)), _ptr(cl),
            dst, counts))
        self.n_labels = list(counts)
        self.n_images = n
        self._uploaded = images         # page-locked sources are read asynchronously (the call ends with a synchronisation)
        return segm

    def _device_array(self, image, which):
        ptr = _vp()
        _check(load_library().imsegm_batch2d_device_ptr(self._h, int(image), which, C.byref(ptr)))
        return DeviceArray(ptr.value, self.shape, '<i4', self)

    def segm_device_array(self, image):
        """the segmentation of image ``image`` of the last batch (int32 H x W) as a device array"""
        return self._device_array(image, 1)

    def labels_device_array(self, image):
        """the superpixel map of image ``image`` of the last batch (int32 H x W) as a device array"""
        return self._device_array(image, 0)

    def get_labels(self, image):
        """superpixel map of image ``image`` of the last batch on the host (int64, the dtype scikit-image leaks)"""
        arr = self.labels_device_array(image)
        out = np.empty(self.shape, dtype=np.int32)
        self.ctx.copy(out.ctypes.data, arr.__cuda_array_interface__['data'][0], out.nbytes, synchronize=True)
        return out.astype(np.int64)


class DeviceArray(object):
    """a result buffer in HBM, exposed through ``__cuda_array_interface__`` (zero-copy hand-over to
    ``torch.as_tensor(..., device='cuda')`` / RCCL on the same HIP runtime)"""

    def __init__(self, ptr, shape, typestr, owner):
        self.owner = owner          # keeps the session alive
        self.__cuda_array_interface__ = {'shape': tuple(shape), 'typestr': typestr, 'data': (int(ptr), False),
                                         'version': 2, 'strides': None}


def gc_grid_fallbacks():
    """grid-wide graph cuts of this process that gave up waiting for a workgroup and were redone by the single workgroup"""
    return int(load_library().imsegm_debug_gc_grid_fallbacks())


def slic_sweep_runs():
    """(2-D SLIC runs whose sweeps after the first ran in the one persistent launch, runs of those that were handed back to the
    per-sweep launches) of this process"""
    a, b = C.c_long(0), C.c_long(0)
    _check(load_library().imsegm_debug_slic_sweep_runs(C.byref(a), C.byref(b)))
    return a.value, b.value


def assign_sweeps_per_launch(max_iter=10):
    """sweeps one launch of the dominant SLIC kernel covered in the runs so far: max_iter - 1 when the persistent kernel took
    them all, 1 with the per-sweep launches (bench.py: algorithmic bytes per launch)"""
    persistent, fallback = slic_sweep_runs()
    return max_iter - 1 if (persistent > 0 and fallback == 0) else 1


def _device_array(sess, which, shape, typestr):
    ptr = _vp()
    _check(load_library().imsegm_image2d_device_ptr(sess._h, which, C.byref(ptr)))
    sess.ctx.synchronize()
    return DeviceArray(ptr.value, shape, typestr, sess)


def segm_device_array(sess):
    """the gathered segmentation (``graph_labels[slic]``, int32 H x W) as a device array"""
    return _device_array(sess, 1, sess.shape, '<i4')


def cut_general_graph(edges, edge_weights, unary_cost, pairwise_cost, n_iter=-1, algorithm='expansion',
                      return_energy=False, ctx=None):
    """drop-in for ``gco.cut_general_graph`` (gco-wrapper), expansion algorithm, on the GPU"""
    if algorithm != 'expansion':
        raise NotImplementedError('only algorithm="expansion" is implemented on the HIP path')
    ctx = ctx or default_context()
    edges = np.ascontiguousarray(np.asarray(edges).reshape(-1, 2), dtype=np.int32)
    ew = np.ascontiguousarray(edge_weights, dtype=np.float64)
    un = np.ascontiguousarray(unary_cost, dtype=np.float64)
    pw = np.ascontiguousarray(pairwise_cost, dtype=np.float64)
    if un.ndim != 2 or pw.shape != (un.shape[1], un.shape[1]) or len(ew) != len(edges):
        raise ValueError('shape mismatch among edges %r, weights %r, unary %r, pairwise %r' %
                         (edges.shape, ew.shape, un.shape, pw.shape))
    labels = np.empty(un.shape[0], dtype=np.int32)
    energy = C.c_int64(0)
    _check(load_library().imsegm_cut_general_graph(ctx._h, _ptr(edges), len(edges), _ptr(ew), _ptr(un), un.shape[0],
                                                   un.shape[1], _ptr(pw), int(n_iter), _ptr(labels), C.byref(energy)))
    return (labels, energy.value) if return_energy else labels


def cut_grid_graph(unary_cost, pairwise_cost, cost_v, cost_h, n_iter=-1, algorithm='expansion', ctx=None):
    """drop-in for ``gco.cut_grid_graph`` (gco-wrapper, reference ``region_growing.py:248``): 4-connected pixel grid with
    vertical / horizontal edge costs ``cost_v`` (H-1 x W) / ``cost_h`` (H x W-1); pygco hands GCO the vertical edges followed
    by the horizontal ones of the general graph, which is what happens here.  Returns int32 labels of the H*W sites."""
    unary_cost = np.asarray(unary_cost, dtype=np.float64)
    if unary_cost.ndim != 3:
        raise ValueError('unary_cost must be height x width x labels')
    height, width, n_labels = unary_cost.shape
    cost_v, cost_h = np.asarray(cost_v, dtype=np.float64), np.asarray(cost_h, dtype=np.float64)
    if cost_v.shape != (height - 1, width) or cost_h.shape != (height, width - 1):
        raise ValueError('cost_v / cost_h do not match the grid %d x %d' % (height, width))
    index = np.arange(height * width, dtype=np.int32).reshape(height, width)
    edges = np.concatenate([np.stack([index[:-1].ravel(), index[1:].ravel()], axis=1),
                            np.stack([index[:, :-1].ravel(), index[:, 1:].ravel()], axis=1)], axis=0)
    weights = np.concatenate([cost_v.ravel(), cost_h.ravel()])
    return cut_general_graph(edges, weights, unary_cost.reshape(height * width, n_labels), pairwise_cost, n_iter=n_iter,
                             algorithm=algorithm, ctx=ctx)


def assume_bg_on_boundary(work, strips, bg_label, ctx=None):
    """``imsegm_assume_bg_on_boundary`` on a contiguous int32 label image (modified in place); returns the label that
    dominates the four border strips"""
    found = C.c_int(0)
    ctx = ctx or default_context()
    _check(load_library().imsegm_assume_bg_on_boundary(ctx._h, _ptr(work), work.shape[0], work.shape[1], _ptr(strips), int(bg_label),
                                                       C.byref(found)))
    return found.value


def label_hist2d(segm, windows, struc_elem, nb_labels, ctx=None):
    """``computeLabelHistogram2d`` for a batch of windows (``imsegm_label_hist2d``): uint32 [P, nb_labels]"""
    ctx = ctx or default_context()
    segm = np.ascontiguousarray(segm, dtype=np.int16)
    selem = np.ascontiguousarray(struc_elem, dtype=np.int16)
    windows = np.ascontiguousarray(windows, dtype=np.int32).reshape(-1, 6)
    out = np.zeros((len(windows), int(nb_labels)), dtype=np.uint32)
    _check(load_library().imsegm_label_hist2d(ctx._h, _ptr(segm), segm.shape[0], segm.shape[1], _ptr(windows), len(windows),
                                              _ptr(selem), selem.shape[0], selem.shape[1], int(nb_labels), _ptr(out)))
    return out


def ray_features_binary2d(seg_binary, positions, directions, edge, ctx=None):
    """``computeRayFeaturesBinary2d`` for a batch of positions (``imsegm_ray_features_binary2d``): float32 [P, A]"""
    ctx = ctx or default_context()
    seg = np.ascontiguousarray(seg_binary, dtype=np.int8)
    positions = np.ascontiguousarray(positions, dtype=np.int32).reshape(-1, 2)
    directions = np.ascontiguousarray(directions, dtype=np.float32).reshape(-1, 2)
    out = np.empty((len(positions), len(directions)), dtype=np.float32)
    _check(load_library().imsegm_ray_features_binary2d(ctx._h, _ptr(seg), seg.shape[0], seg.shape[1], _ptr(positions), len(positions),
                                                       _ptr(directions), len(directions), int(edge), _ptr(out)))
    return out
