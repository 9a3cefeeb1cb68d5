"""Shared set-up of the golden generators that run the reference UNCHANGED from /root/reference under the build
container's conda Python 3.9 (real scikit-image 0.18.3, scikit-learn, Cython): see make_golden_reference.py for the
full description.  Build container only -- nothing here is imported by the tests."""
import ctypes as C
import os
import subprocess
import sys
import sysconfig
import tempfile
import types
import warnings
import zlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = '/root/reference'
warnings.filterwarnings('ignore')


def crc(arr):
    return zlib.crc32(np.ascontiguousarray(arr).tobytes())


def build_cython(tmp):
    import numpy
    pyx = os.path.join(REF, 'imsegm', 'features_cython.pyx')
    cpp = os.path.join(tmp, 'features_cython.cpp')
    target = os.path.join(tmp, 'features_cython' + sysconfig.get_config_var('EXT_SUFFIX'))
    subprocess.check_call([sys.executable, '-m', 'cython', '--cplus', '-3', pyx, '-o', cpp])
    subprocess.check_call(['g++', '-shared', '-fPIC', '-O3', '-ffast-math', '-march=x86-64-v2', '-w',
                           '-DNPY_NO_DEPRECATED_API=NPY_1_7_API_VERSION', '-I' + numpy.get_include(),
                           '-I' + sysconfig.get_paths()['include'], cpp, '-o', target])
    return target


class ReferenceEnv(object):
    """`with ReferenceEnv() as env:` -> env.pipelines / .descriptors / .graph_cuts / .superpixels are the reference's
    own modules; env.recorded holds the inputs of the last `gco.cut_general_graph` call (bridged to the oracle: gco
    exists nowhere in the container)"""

    def __enter__(self):
        self.recorded = recorded = {}
        lib = C.CDLL(os.path.join(ROOT, 'oracle', 'liboracle.so'))
        lib.orc_cut_general_graph.restype = C.c_int64

        def bridge_cut_general_graph(edges, edge_weights, unary_cost, pairwise_cost, n_iter=-1, algorithm='expansion', **kw):
            edges = np.ascontiguousarray(edges, dtype=np.int32)
            weights = np.ascontiguousarray(edge_weights, dtype=np.float64)
            unary = np.ascontiguousarray(unary_cost, dtype=np.float64)
            pairwise = np.ascontiguousarray(pairwise_cost, dtype=np.float64)
            recorded.update(edges=edges.copy(), edge_weights=weights.copy(), unary=unary.copy(), pairwise=pairwise.copy())
            labels = np.zeros(unary.shape[0], dtype=np.int32)
            ptr = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
            lib.orc_cut_general_graph(ptr(edges), C.c_int(len(edges)), ptr(weights), ptr(unary), C.c_int(unary.shape[0]),
                                      C.c_int(unary.shape[1]), ptr(pairwise), C.c_int(n_iter), ptr(labels))
            return labels

        for name in ('nibabel', 'planar', 'gco', 'OleFileIO_PL'):
            sys.modules[name] = types.ModuleType(name)
        sys.modules['planar'].line = types.ModuleType('planar.line')
        sys.modules['gco'].cut_general_graph = bridge_cut_general_graph
        sys.modules['gco'].cut_grid_graph = bridge_cut_general_graph
        sys.path.insert(0, REF)
        sys.path.insert(1, ROOT)
        import imsegm
        self._tmp = tempfile.TemporaryDirectory()
        build_cython(self._tmp.name)
        imsegm.__path__.append(self._tmp.name)
        import imsegm.descriptors as seg_fts
        import imsegm.graph_cuts as seg_gc
        import imsegm.pipelines as seg_pipe
        import imsegm.superpixels as seg_spx
        assert seg_fts.USE_CYTHON, 'the reference must run its Cython descriptor path'
        import skimage
        import sklearn
        self.descriptors, self.graph_cuts, self.pipelines, self.superpixels = seg_fts, seg_gc, seg_pipe, seg_spx
        self.versions = np.array('scikit-image %s, scikit-learn %s, numpy %s' % (skimage.__version__, sklearn.__version__, np.__version__))
        return self

    def __exit__(self, *exc):
        self._tmp.cleanup()
        return False


def model_arrays(model, prefix=''):
    """parameters of the reference's `Pipeline([StandardScaler, GaussianMixture])` as plain arrays"""
    scaler, gmm = model.steps[0][1], model.steps[-1][1]
    return {prefix + 'scaler_mean': scaler.mean_, prefix + 'scaler_scale': scaler.scale_, prefix + 'gmm_weights': gmm.weights_,
            prefix + 'gmm_means': gmm.means_, prefix + 'gmm_covariances': gmm.covariances_,
            prefix + 'gmm_precisions_cholesky': gmm.precisions_cholesky_}
