"""TEST-LOGIC DRY RUN (test infrastructure, opt-in, never loaded by default and never by the product).

    python -m pytest -p dryrun_plugin tests/test_gpu_zz_reference.py -m gpu        # from the tests/ directory on sys.path:
    PYTHONPATH=tests python -m pytest -p dryrun_plugin tests -m gpu -k "zz or api"

The `-m gpu` tests need an MI355X.  GPU time is scarce, and a typo in a new GPU test (a wrong tolerance, argument order,
fixture name) would only show up there.  This pytest plugin replaces the ctypes session classes of `pyimsegm_amd._hip`
with stand-ins that answer from the CPU oracle, so that the PYTHON side of the GPU tests, of `bench.py` and of the host
code (pipelines, helper processes, threading) can be exercised on a machine without a GPU.  It says NOTHING about the
kernels -- a dry run passing here proves only that the test logic is sound, given that the HIP path equals the oracle
(which the real `-m gpu` run establishes).  The product has no CPU fallback: `pyimsegm_amd` never imports this file or
`oracle/`, and without the HIP library its calls raise.  Texture (`lm_*`) entry points are not emulated.
"""
import sys
import numpy as np
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as orc
from pyimsegm_amd import _hip


class FakeCtx(object):
    users = 0
    def __init__(self): self.idle_sessions = {}; self._h = 1
    def synchronize(self): pass
    def close(self): pass

_CTX = FakeCtx()


class Image2D(object):
    def __init__(self, height, width, ctx=None):
        self.ctx = _CTX; self.shape = (int(height), int(width)); self.n_labels = 0; self.img = None; self.labels = None
    def close(self): pass
    def upload(self, image):
        image = np.asarray(image)
        assert image.ndim == 3 and image.shape[2] == 3 and image.shape[:2] == self.shape
        if image.dtype not in (np.uint8, np.float32, np.float64): image = image.astype(np.float64)
        self.img = image; return self
    def slic(self, n_segments, compactness, sigma=1., normalize=2, max_iter=10, enforce_connectivity=True,
             min_size_factor=0.5, max_size_factor=3., start_label=0, max_candidates=0, slic_zero=False):
        img = self.img
        norm = None
        if normalize == 1 or (normalize == 2 and (img.min() != 0. or img.max() != 1.)):
            norm = (float(img.min()), float(img.max()))
        lab = orc.slic(img, n_segments, compactness, sigma=sigma, max_iter=max_iter, enforce_connectivity=enforce_connectivity,
                       min_size_factor=min_size_factor, max_size_factor=max_size_factor, start_label=start_label,
                       normalize=norm, slic_zero=slic_zero)
        self.labels = np.asarray(lab, dtype=np.int32); self.n_labels = int(self.labels.max()) + 1
        return self.n_labels
    def get_labels(self): return self.labels.astype(np.int64)
    def get_labels_int32(self): return self.labels.astype(np.int32)
    def set_labels(self, labels, n_labels=None):
        labels = np.ascontiguousarray(labels, dtype=np.int32); assert labels.shape == self.shape
        self.labels = labels; self.n_labels = int(labels.max()) + 1 if n_labels is None else int(n_labels); return self
    def label_hist(self, annot, nb_annot=None):
        annot = np.asarray(annot, dtype=np.int32)
        nb = int(annot.max()) + 1 if nb_annot is None else int(nb_annot)
        out = np.zeros((self.n_labels, nb), dtype=np.int64)
        np.add.at(out, (self.labels.ravel(), annot.ravel()), 1)
        return out
    def color_stats(self, mean=True, energy=True, var=True):
        img32 = np.asarray(self.img, dtype=np.float32)
        m = orc.color2d_mean(img32, self.labels)
        e = orc.color2d_energy(img32, self.labels) if energy else None
        v = orc.color2d_variance(img32, self.labels, m.astype(np.float32)) if var else None
        return (m if mean else None), e, v
    def graph(self):
        vertices, edges = orc.adjacency(self.labels)
        centres = np.asarray(orc.centers(self.labels), dtype=np.float64).reshape(self.n_labels, -1)
        present = np.zeros(self.n_labels, dtype=bool); present[np.asarray(vertices)] = True
        return np.array(edges, dtype=np.int32).reshape(-1, 2), centres, present
    def graph_prepare(self): pass
    def gather(self, graph_labels=None, proba=None, to_host=True, segm_out=None):
        segm = np.asarray(graph_labels, dtype=np.int32)[self.labels] if graph_labels is not None else None
        if segm is not None and segm_out is not None: segm_out[...] = segm; segm = segm_out
        self.last_segm = segm
        soft = np.asarray(proba, dtype=np.float64)[self.labels] if proba is not None else None
        return segm, soft
    def run_color(self, image, n_segments, compactness, gmm, pairwise, edge_type='model', feature_flags=(True, True, True),
                  sigma=1., normalize=2, max_iter=10, start_label=0, slic_zero=False, edge_cost=1., use_graphcut=True,
                  classes=None, want_soft=False, pinned=True):
        self.upload(image)
        self.slic(n_segments, compactness, sigma=sigma, normalize=normalize, max_iter=max_iter, start_label=start_label, slic_zero=slic_zero)
        self.features_color(*feature_flags, to_host=False)
        out = self.segment(pairwise, edge_type, edge_cost, gmm=gmm, use_graphcut=use_graphcut, classes=classes, want_soft=want_soft)
        return out['segm'], out.get('soft')
    def features_color(self, mean=True, std=True, energy=True, to_host=True):
        m, e, v = self.color_stats(True, energy, std)
        blocks = ([m] if mean else []) + ([np.sqrt(v)] if std else []) + ([e] if energy else [])
        fts = np.nan_to_num(np.hstack(blocks)); fts[fts == 0] = 0
        place = getattr(self, '_place', None)
        self._place = None
        if place is not None and place[0] != fts.shape[1]:       # a group of a wider table
            if getattr(self, 'features', None) is None or self.features.shape[1] != place[0]:
                self.features = np.zeros((fts.shape[0], place[0]))
            self.features[:, place[1]:place[1] + fts.shape[1]] = fts
            return None
        self.features = fts
        return fts if to_host else None
    def features_place(self, total_columns, column):
        self._place = (int(total_columns), int(column))
        return self
    def get_features(self, columns):
        assert self.features.shape[1] == columns
        return np.array(self.features)
    def segment(self, pairwise, edge_type='model', edge_cost=1., gmm=None, proba=None, use_graphcut=True, classes=None,
                want_segm=True, want_soft=False, want_graph_labels=False, want_proba=False, debug=False, pinned=True,
                keep_soft_on_device=False, segm_dtype=None, soft_dtype=None):
        """the fused call, restated with the host mirror functions of graph_cuts + the oracle's alpha-expansion"""
        import pyimsegm_amd.graph_cuts as G
        from scipy.special import logsumexp
        if gmm is not None:
            x = np.array(self.features, dtype=np.float64)
            if gmm.scaler_mean is not None: x = x - gmm.scaler_mean
            if gmm.scaler_scale is not None: x = x / gmm.scaler_scale
            lp = np.stack([np.sum((x @ pc - mp)**2, axis=1) for pc, mp in zip(gmm.prec_chol, gmm.mu_proj)], axis=1)
            wl = -0.5 * (gmm.const_term + lp) + gmm.log_det + gmm.log_weights
            proba = np.exp(wl - logsumexp(wl, axis=1)[:, None])
        proba = np.asarray(proba, dtype=np.float64)[:self.n_labels]
        edges, centres, _ = self.graph()
        weights = G.edge_weights_from_graph(edges, centres, getattr(self, 'features', None), proba, edge_type) * edge_cost
        unary = G.compute_unary_cost(proba)
        pairwise = np.asarray(pairwise, dtype=np.float64)
        if use_graphcut:
            gl = np.asarray(orc.cut_general_graph(edges, weights, unary, pairwise, n_iter=-1), dtype=np.int32)
        else:
            gl = np.argmin(unary, axis=-1).astype(np.int32)
        lut = gl if classes is None else np.asarray(classes, dtype=np.int32)[gl]
        out = {}
        self.last_segm = lut[self.labels]
        if want_segm: out['segm'] = self.last_segm if segm_dtype is None else self.last_segm.astype(segm_dtype)
        if want_soft: out['soft'] = proba[self.labels] if soft_dtype is None else proba[self.labels].astype(soft_dtype)
        if want_graph_labels or debug: out['graph_labels'] = gl
        if want_proba or debug: out['proba'] = proba
        if debug: out.update(edges=edges, edge_weights=weights, unary=unary, centres=centres)
        return out


class Batch2D(object):
    """stand-in of _hip.Batch2D: the images of a batch answered one after the other by the oracle-backed Image2D above"""
    def __init__(self, max_images, height, width, ctx=None):
        self.ctx = _CTX; self.shape = (int(height), int(width)); self.max_images = int(max_images); self.n_labels = []; self.n_images = 0
        self.sessions = []
    def close(self): pass
    def run_color(self, images, n_segments, compactness, gmm, pairwise, edge_type='model', feature_flags=(True, True, True),
                  sigma=1., normalize=2, max_iter=10, start_label=0, edge_cost=1., use_graphcut=True, classes=None, to_host=True,
                  pinned=True, out=None):
        assert 0 < len(images) <= self.max_images
        self.sessions, segm = [], []
        for k, image in enumerate(images):
            sess = Image2D(*self.shape)
            got, _ = sess.run_color(image, n_segments, compactness, gmm, pairwise, edge_type, feature_flags, sigma, normalize, max_iter,
                                    start_label, False, edge_cost, use_graphcut, classes)
            if out is not None:
                out[k][...] = got
                got = out[k]
            self.sessions.append(sess); segm.append(got)
        self.n_labels = [s.n_labels for s in self.sessions]; self.n_images = len(images)
        return segm if to_host else None
    def segm_device_array(self, image): return self.sessions[image].last_segm
    def labels_device_array(self, image): return self.sessions[image].labels
    def get_labels(self, image): return self.sessions[image].labels.astype(np.int64)


class Volume3D(Image2D):
    def __init__(self, depth, height, width, ctx=None):
        self.ctx = _CTX; self.shape = (int(depth), int(height), int(width)); self.n_labels = 0
    def upload(self, volume):
        volume = np.asarray(volume); assert volume.shape == self.shape; self.img = volume; self.dtype = volume.dtype; return self
    def all_finite(self):
        return bool(np.isfinite(self.img).all())
    def graph(self):
        vertices, edges = orc.adjacency(self.labels)
        centres = np.asarray(orc.centers(self.labels), dtype=np.float64).reshape(self.n_labels, -1)
        present = np.zeros(self.n_labels, dtype=bool); present[np.asarray(vertices)] = True
        return np.array(edges, dtype=np.int32).reshape(-1, 2), centres, present
    def slic(self, n_segments, compactness, sigma=1., spacing=(1., 1., 1.), max_iter=10, enforce_connectivity=True,
             min_size_factor=0.5, max_size_factor=3., start_label=0):
        if self.img.dtype == np.float32:
            lab = orc.slic_gray3d_float32(self.img, n_segments, compactness, sigma=sigma, spacing=spacing, max_iter=max_iter,
                                          enforce_connectivity=enforce_connectivity, start_label=start_label)
        else:
            lab = orc.slic(self.img, n_segments, compactness, sigma=sigma, spacing=spacing, multichannel=False, max_iter=max_iter,
                           enforce_connectivity=enforce_connectivity, start_label=start_label)
        self.labels = np.asarray(lab, dtype=np.int32); self.n_labels = int(self.labels.max()) + 1; return self.n_labels
    def label_cc(self):
        self.labels = orc.label_cc(self.labels).astype(np.int32); self.n_labels = int(self.labels.max()) + 1; return self.n_labels
    def gray_stats(self, mean=True, energy=True, var=True):
        v32 = np.asarray(self.img, dtype=np.float32)
        m = orc.gray3d_stat(v32, self.labels, 'mean')
        e = orc.gray3d_stat(v32, self.labels, 'energy') if energy else None
        v = orc.gray3d_stat(v32, self.labels, 'var', m.astype(np.float32)) if var else None
        return (m if mean else None), e, v


def _assume_bg_on_boundary(work, strips, bg_label, ctx=None):
    """numpy stand-in of imsegm_assume_bg_on_boundary (border histogram over the four strips, label exchange in place)"""
    parts = [work[strips[4 * q]:strips[4 * q + 1], strips[4 * q + 2]:strips[4 * q + 3]].ravel() for q in range(4)]
    found = int(np.argmax(np.bincount(np.concatenate(parts))))
    if found != bg_label:
        a, b = work == found, work == bg_label
        work[a], work[b] = bg_label, found
    return found


def pytest_configure(config):
    _hip.assume_bg_on_boundary = _assume_bg_on_boundary
    _hip.Image2D = Image2D
    _hip.Batch2D = Batch2D
    _hip.Volume3D = Volume3D
    _hip.default_context = lambda: _CTX
    _hip.cut_general_graph = lambda e, w, u, p, n_iter=-1, algorithm='expansion', **k: orc.cut_general_graph(
        np.asarray(e, dtype=np.int32).reshape(-1, 2), w, u, p, n_iter=n_iter)
    _hip.segm_device_array = lambda sess: sess.last_segm          # (the real one exposes the HBM buffer)
    _hip.pinned_empty = lambda shape, dtype: np.empty(shape, dtype)
    import pyimsegm_amd.graph_cuts as G
    G._hip = _hip


def _ctx_extra():
    FakeCtx.profile_enable = lambda self, enable=True: None
    FakeCtx.profile_reset = lambda self: None
    FakeCtx.profile_get = lambda self, group: (1.0, 10)

_ctx_extra()
