"""Wall time of the REFERENCE ITSELF on the benchmark image (build container only: needs /root/reference and the conda
Python 3.9 with scikit-image; see tests/golden/make_golden_reference.py for the stubs and the gco bridge):

    /opt/conda/bin/python3.9 tools/time_reference.py

Measured there (1 core of an 8-vCPU Xeon @ 2.1 GHz): 7.8 s per 2048 x 2048 image with the Cython descriptors
(0.535 Mpixels/s), 33.2 s with the numpy descriptors the driver selects (USE_CYTHON = False, 0.126 Mpixels/s).
"""
import sys, time, types, warnings, tempfile, os
warnings.filterwarnings('ignore')
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
import make_golden_reference as m
import ctypes as C
for name in ('nibabel', 'planar', 'gco', 'OleFileIO_PL'):
    sys.modules[name] = types.ModuleType(name)
sys.modules['planar'].line = types.ModuleType('planar.line')
lib = C.CDLL(os.path.join(ROOT, 'oracle', 'liboracle.so')); lib.orc_cut_general_graph.restype = C.c_int64
def bridge(edges, edge_weights, unary_cost, pairwise_cost, n_iter=-1, algorithm='expansion', **kw):
    e = np.ascontiguousarray(edges, np.int32); w = np.ascontiguousarray(edge_weights, np.float64)
    u = np.ascontiguousarray(unary_cost, np.float64); p = np.ascontiguousarray(pairwise_cost, np.float64)
    out = np.zeros(u.shape[0], np.int32); ptr = lambda a: a.ctypes.data_as(C.c_void_p)
    lib.orc_cut_general_graph(ptr(e), C.c_int(len(e)), ptr(w), ptr(u), C.c_int(u.shape[0]), C.c_int(u.shape[1]), ptr(p), C.c_int(n_iter), ptr(out))
    return out
sys.modules['gco'].cut_general_graph = bridge; sys.modules['gco'].cut_grid_graph = bridge
sys.path.insert(0, '/root/reference')
import imsegm
tmp = tempfile.mkdtemp(); m.build_cython(tmp); imsegm.__path__.append(tmp)
import imsegm.descriptors as seg_fts, imsegm.pipelines as seg_pipe
sys.path.insert(0, ROOT)
from pyimsegm_amd.utilities.synthetic import voronoi_image
img = voronoi_image(2048, 2048, seed=1)
feats = {'color': ('mean', 'std', 'energy')}
for use_cython in (True, False):
    seg_fts.USE_CYTHON = use_cython
    np.random.seed(0)
    t0 = time.perf_counter()
    model, _ = seg_pipe.estim_model_classes_group([img], 3, feats, sp_size=46, sp_regul=0.2, nb_workers=1)
    t1 = time.perf_counter()
    segm, soft = seg_pipe.segment_color2d_slic_features_model_graphcut(img, model, feats, sp_size=46, sp_regul=0.2, gc_regul=2.0, gc_edge_type='model')
    t2 = time.perf_counter()
    print('USE_CYTHON=%s: model group (slic+fts+fit) %.2f s, segment %.2f s -> %.3f Mpx/s; classes %s' % (use_cython, t1 - t0, t2 - t1, img.shape[0]*img.shape[1]/(t2-t1)/1e6, np.bincount(segm.ravel()).tolist()))
