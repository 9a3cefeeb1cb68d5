"""Timing of the gray-volume (supervoxel) path on one GPU (BASELINE config 5 is 64 x 4096 x 4096).

    python tools/bench_volume.py [D H W] [sp_size] [--pipeline] [--uint8]

The synthetic volume is float32 by default; note that scikit-image 0.18 runs a float32 volume in
float32 while this path widens it to float64.  `--uint8` quantises the same volume to uint8, for
which the path reproduces scikit-image bit for bit.
"""
import sys
import time

import numpy as np

sys.path.insert(0, __file__.rsplit('/', 2)[0])
from pyimsegm_amd import _hip, pipelines  # noqa: E402
from pyimsegm_amd.superpixels import _slic3d_params  # noqa: E402


def volume(shape, seed=5):
    """three nested ellipsoids + N(0, 0.05) noise, generated slice by slice"""
    rng = np.random.default_rng(seed)
    zs, ys, xs = [np.linspace(-1, 1, s, dtype=np.float32) for s in shape]
    ryx = (ys[:, None] / 0.7)**2 + (xs[None, :] / 0.8)**2
    vol = np.empty(shape, dtype=np.float32)
    for i, z in enumerate(zs):
        r = np.sqrt((z / 0.9)**2 + ryx)
        vol[i] = (r < 0.9) * np.float32(0.3) + (r < 0.6) * np.float32(0.3) + (r < 0.3) * np.float32(0.3)
        vol[i] += np.float32(0.05) * rng.standard_normal(r.shape, dtype=np.float32)
    return vol


def main():
    args = [a for a in sys.argv[1:] if not a.startswith('--')]
    shape = tuple(int(v) for v in args[0:3]) if len(args) >= 3 else (64, 512, 512)
    sp = int(args[3]) if len(args) > 3 else 15
    vol = volume(shape)
    if '--uint8' in sys.argv:
        np.clip(vol, 0, 1, out=vol)
        vol = (vol * np.float32(255)).astype(np.uint8)
    n = vol.size
    n_seg, compact = _slic3d_params(shape, sp, 0.2, (1, 1, 1))
    print('volume %r, %d voxels, n_segments %d, compactness %d' % (shape, n, n_seg, compact))
    ctx = _hip.default_context()
    sess = _hip.Volume3D(*shape).upload(vol)
    for rep in range(2):
        ctx.synchronize()
        t0 = time.perf_counter()
        k0 = sess.slic(n_seg, compact, sigma=1., spacing=(1, 1, 1))
        ctx.synchronize()
        t1 = time.perf_counter()
        k = sess.label_cc()
        ctx.synchronize()
        t2 = time.perf_counter()
        mean, energy, var = sess.gray_stats()
        t3 = time.perf_counter()
        edges, centres, present = sess.graph()
        t4 = time.perf_counter()
        print('run %d: slic %.1f ms (%d labels) | label_cc %.1f ms (%d) | gray stats %.1f ms | graph %.1f ms (%d edges) | '
              '%.1f Mvoxel/s through SLIC' % (rep, (t1 - t0) * 1e3, k0, (t2 - t1) * 1e3, k, (t3 - t2) * 1e3,
                                              (t4 - t3) * 1e3, len(edges), n / (t1 - t0) / 1e6))
    if n <= 2**24 or '--pipeline' in sys.argv:
        # the stages of pipe_gray3d_slic_features_model_graphcut (pipelines.py:382-431) with a clock on each
        from pyimsegm_amd import graph_cuts as gc
        from pyimsegm_amd.descriptors import compute_selected_features_gray3d, norm_features
        from pyimsegm_amd.pipelines import _ShapeOnly
        from pyimsegm_amd.superpixels import _open_volume, _run_slic3d
        marks = [('start', time.perf_counter())]

        def mark(name):
            ctx.synchronize()
            marks.append((name, time.perf_counter()))

        np.random.seed(0)
        sess2 = _open_volume(vol); mark('upload')
        _run_slic3d(sess2, sp, 0.2, (1, 1, 1)); mark('slic + label')
        feats, _ = compute_selected_features_gray3d(vol, _ShapeOnly(sess2.shape), {'color': ['mean', 'std', 'energy']}, sess=sess2)
        feats[np.isnan(feats)] = 0
        feats, _ = norm_features(feats); mark('features')
        model = gc.estim_class_model(feats, 3); mark('GMM fit (host)')
        proba = model.predict_proba(feats); mark('predict_proba (host)')
        edges, weights = gc.compute_edge_weights(_ShapeOnly(sess2.shape), vol, feats, proba, 'model', _session=sess2); mark('graph + edge weights')
        unary = gc.compute_unary_cost(proba)
        pairwise = gc.compute_pairwise_cost(0.1, proba.shape)
        labels = gc.cut_general_graph(edges, weights, unary, pairwise, n_iter=-1); mark('graph cut')
        segm, _ = sess2.gather(labels); mark('gather + D2H')
        print('pipeline stages: ' + ' | '.join('%s %.0f ms' % (b[0], (b[1] - a[1]) * 1e3) for a, b in zip(marks, marks[1:])) +
              ' | total %.0f ms; K %d, E %d, classes %r' % ((marks[-1][1] - marks[0][1]) * 1e3, len(feats), len(edges), np.bincount(segm.ravel())))
        return
    if n <= 2**24 or '--pipeline' in sys.argv:
        t0 = time.perf_counter()
        np.random.seed(0)
        segm = pipelines.pipe_gray3d_slic_features_model_graphcut(vol, 3, {'color': ['mean', 'std', 'energy']},
                                                                  spacing=(1, 1, 1), sp_size=sp, sp_regul=0.2, gc_regul=0.1)
        t1 = time.perf_counter()
        print('pipe_gray3d_slic_features_model_graphcut: %.1f ms, classes %r' % ((t1 - t0) * 1e3, np.bincount(segm.ravel())))


if __name__ == '__main__':
    main()
