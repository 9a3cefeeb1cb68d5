#!/usr/bin/env python
"""Golden vectors AT THE SIZE of BASELINE configs 3, 4 and 5 from the reference's own code (build container only):

    /opt/conda/bin/python3.9 tests/golden/make_golden_configs.py c3        # ~15 min of scipy on one core
    /opt/conda/bin/python3.9 tests/golden/make_golden_configs.py c4        # ~5 min
    /opt/conda/bin/python3.9 tests/golden/make_golden_configs.py c5mid     # ~2 min
    /opt/conda/bin/python3.9 tests/golden/make_golden_configs.py c5full    # ~1 h, ~45 GB of host memory

The reference runs UNCHANGED from /root/reference with its real dependencies (scikit-image 0.18.3, scikit-learn, scipy,
its compiled features_cython.pyx; `gco.cut_general_graph` -- absent everywhere -- bridged to the oracle, so the
`segm` checksums are independent evidence for SLIC + descriptors + class model + terms and self-referential for the
cut: tests/golden/_reference_env.py).  Label maps are stored as CRC32, descriptors and class models with all their
numbers.  `bench.py --config N` and tests/test_gpu_zz_configs.py compare the HIP path with these files.

  c3      configs[2]: the 2048 x 2048 benchmark image, {'tLM': ('mean', 'std', 'energy')} (76 kernels, F = 180), stage by
          stage in the order of imsegm/pipelines.py:86-110               -> reference_c3.npz
  c4      configs[3]: 64 images 647 x 1024 (seeds 100..163), run_segm_slic_model_graphcut.py:476-514 (group model fitted
          over all images, then every image segmented with it)           -> reference_c4.npz
  c5mid   configs[4] on a 32 x 512 x 512 float32 volume (many bricks on every axis): imsegm/pipelines.py:382-431 stage by
          stage                                                          -> reference_c5.npz
  c5full  the full 64 x 4096 x 4096 float32 volume through imsegm/superpixels.py:72-112 (real scikit-image slic +
          measure.label): CRC32 of the supervoxel map                    -> reference_c5_full.npz
"""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from _reference_env import ROOT, ReferenceEnv, crc, model_arrays  # noqa: E402

sys.path.insert(1, ROOT)
FEATURES_SET_COLOR = {'color': ('mean', 'std', 'energy')}
FEATURES_SET_LM = {'tLM': ('mean', 'std', 'energy')}
C2_IMAGE = dict(height=2048, width=2048, seed=1)
C2_PARAMS = dict(sp_size=46, sp_regul=0.2, nb_classes=3, gc_regul=2.0, gc_edge_type='model')
C4_SHAPE, C4_SEEDS = (647, 1024), list(range(100, 164))
C4_PARAMS = dict(sp_size=35, sp_regul=0.2, nb_classes=3, gc_regul=2.0, gc_edge_type='model')
C5_PARAMS = dict(sp_size=15, sp_regul=0.2, spacing=(1, 1, 1), nb_classes=3, gc_regul=0.1)
C5_MID_SHAPE = (32, 512, 512)
C5_FULL_SHAPE = (64, 4096, 4096)


def config3():
    from pyimsegm_amd.utilities.synthetic import voronoi_image
    p = C2_PARAMS
    with ReferenceEnv() as env:
        image = voronoi_image(C2_IMAGE['height'], C2_IMAGE['width'], seed=C2_IMAGE['seed'])
        t0 = time.time()
        slic, features = env.pipelines.compute_color2d_superpixels_features(image, FEATURES_SET_LM, sp_size=p['sp_size'],
                                                                            sp_regul=p['sp_regul'])
        t_fts = time.time() - t0
        _, names = env.descriptors.compute_selected_features_img2d(image[:64, :64], np.zeros((64, 64), dtype=int), FEATURES_SET_LM)
        np.random.seed(0)
        model = env.graph_cuts.estim_class_model(features, p['nb_classes'], 'GMM', None, True)
        proba = model.predict_proba(features)
        graph_labels = env.graph_cuts.segment_graph_cut_general(slic, proba, image, features, p['gc_regul'], p['gc_edge_type'])
        segm = graph_labels[slic]
        out = {'versions': env.versions, 'image_crc': np.array(crc(image), dtype=np.uint32),
               'slic_crc': np.array(crc(np.asarray(slic).astype(np.int32)), dtype=np.uint32),
               'nb_superpixels': np.array(int(slic.max()) + 1), 'features': np.asarray(features, dtype=np.float64),
               'names': np.array(names), 'proba': proba, 'graph_labels': np.asarray(graph_labels).astype(np.int32),
               'segm_crc': np.array(crc(np.asarray(segm).astype(np.int32)), dtype=np.uint32),
               'class_counts': np.bincount(np.asarray(segm).ravel()), 'nb_edges': np.array(len(env.recorded['edges'])),
               'gc_unary': env.recorded['unary'], 'gc_edge_weights': env.recorded['edge_weights'],
               'seconds_slic_and_descriptors_one_core': np.array(t_fts)}
        out.update(model_arrays(model))
        np.savez_compressed(os.path.join(HERE, 'reference_c3.npz'), **out)
        print('config 3: K = %d, F = %d, E = %d, classes %s, SLIC + descriptors %.0f s' %
              (out['nb_superpixels'], features.shape[1], out['nb_edges'], out['class_counts'].tolist(), t_fts))


def config4():
    from pyimsegm_amd.utilities.synthetic import voronoi_image
    p = C4_PARAMS
    with ReferenceEnv() as env:
        images = [voronoi_image(C4_SHAPE[0], C4_SHAPE[1], seed=s) for s in C4_SEEDS]
        np.random.seed(0)
        t0 = time.time()
        model, list_features = env.pipelines.estim_model_classes_group(images, p['nb_classes'], FEATURES_SET_COLOR,
                                                                       sp_size=p['sp_size'], sp_regul=p['sp_regul'], nb_workers=1)
        t_group = time.time() - t0
        rows = []
        t0 = time.time()
        for image, fts in zip(images, list_features):
            env.recorded.clear()
            segm, _ = env.pipelines.segment_color2d_slic_features_model_graphcut(
                image, model, FEATURES_SET_COLOR, sp_size=p['sp_size'], sp_regul=p['sp_regul'], gc_regul=p['gc_regul'],
                gc_edge_type=p['gc_edge_type'])
            rows.append((crc(image), crc(np.asarray(segm).astype(np.int32)), len(fts), len(env.recorded['edges']),
                         np.bincount(np.asarray(segm).ravel(), minlength=p['nb_classes'])))
        t_segm = time.time() - t0
        slic_crcs = [crc(np.asarray(env.superpixels.segment_slic_img2d(im, p['sp_size'], p['sp_regul'])).astype(np.int32))
                     for im in images]
        out = {'versions': env.versions, 'seeds': np.array(C4_SEEDS), 'image_crc': np.array([r[0] for r in rows], dtype=np.uint32),
               'slic_crc': np.array(slic_crcs, dtype=np.uint32), 'segm_crc': np.array([r[1] for r in rows], dtype=np.uint32),
               'nb_superpixels': np.array([r[2] for r in rows]), 'nb_edges': np.array([r[3] for r in rows]),
               'class_counts': np.array([r[4] for r in rows]),
               'features_offsets': np.cumsum([0] + [len(f) for f in list_features]),
               'features': np.vstack(list_features).astype(np.float64),
               'seconds_one_core': np.array([t_group, t_segm])}
        out.update(model_arrays(model))
        np.savez_compressed(os.path.join(HERE, 'reference_c4.npz'), **out)
        print('config 4: %d images, K = %d..%d, group model %.0f s, segmentation %.0f s (one core)' %
              (len(images), out['nb_superpixels'].min(), out['nb_superpixels'].max(), t_group, t_segm))


def config5_mid():
    from pyimsegm_amd.utilities.synthetic import config5_volume
    p = C5_PARAMS
    with ReferenceEnv() as env:
        vol = config5_volume(C5_MID_SHAPE)
        t0 = time.time()
        slic = env.superpixels.segment_slic_img3d_gray(vol, sp_size=p['sp_size'], relative_compact=p['sp_regul'], space=p['spacing'])
        t_slic = time.time() - t0
        features, names = env.descriptors.compute_selected_features_gray3d(vol, slic, FEATURES_SET_COLOR)
        features[np.isnan(features)] = 0
        normed, _ = env.descriptors.norm_features(features)          # (no copy: the memory layout reaches the k-means initialisation)
        np.random.seed(0)
        model = env.graph_cuts.estim_class_model(normed, p['nb_classes'])
        proba = model.predict_proba(normed)
        env.recorded.clear()
        graph_labels = env.graph_cuts.segment_graph_cut_general(slic, proba, vol, normed, p['gc_regul'])
        segm = graph_labels[slic]
        np.random.seed(0)
        t0 = time.time()
        segm_pipe = env.pipelines.pipe_gray3d_slic_features_model_graphcut(vol, p['nb_classes'], FEATURES_SET_COLOR, spacing=p['spacing'],
                                                                           sp_size=p['sp_size'], sp_regul=p['sp_regul'], gc_regul=p['gc_regul'])
        t_pipe = time.time() - t0
        assert np.array_equal(segm_pipe, segm)
        out = {'versions': env.versions, 'shape': np.array(C5_MID_SHAPE), 'volume_crc': np.array(crc(vol), dtype=np.uint32),
               'slic_crc': np.array(crc(np.asarray(slic).astype(np.int32)), dtype=np.uint32),
               'nb_supervoxels': np.array(int(slic.max()) + 1), 'features': np.asarray(features, dtype=np.float64),
               'normed': normed, 'proba': proba, 'graph_labels': np.asarray(graph_labels).astype(np.int32),
               'edges_crc': np.array(crc(env.recorded['edges']), dtype=np.uint32), 'nb_edges': np.array(len(env.recorded['edges'])),
               'segm_crc': np.array(crc(np.asarray(segm).astype(np.int32)), dtype=np.uint32),
               'class_counts': np.bincount(np.asarray(segm).ravel()),
               'seconds_one_core': np.array([t_slic, t_pipe])}
        out.update(model_arrays(model))
        np.savez_compressed(os.path.join(HERE, 'reference_c5.npz'), **out)
        print('config 5 (%s): K = %d, E = %d, classes %s; slic + label %.0f s, whole pipeline %.0f s (one core)' %
              (C5_MID_SHAPE, out['nb_supervoxels'], out['nb_edges'], out['class_counts'].tolist(), t_slic, t_pipe))


def config5_full():
    """only the third-party leg (imsegm/superpixels.py:87-112 restated parameter for parameter, as make_golden_skimage.py
    does): the reference's pure-Python graph / centre loops are infeasible on 10^9 voxels"""
    import skimage
    from skimage import measure
    from skimage.segmentation import slic
    from pyimsegm_amd.utilities.synthetic import config5_volume
    p = C5_PARAMS
    t0 = time.time()
    vol = config5_volume(C5_FULL_SHAPE)
    t_gen = time.time() - t0
    vol_crc = crc(vol)
    space = p['spacing']
    nb_pixels = np.prod(vol.shape)
    sp_vol = np.prod(p['sp_size'] / np.asarray(space, dtype=np.float32) * min(space))
    n_seg = int(nb_pixels / sp_vol)
    compact = int((sp_vol * p['sp_regul'])**1.5)
    print('volume generated in %.0f s; n_segments %d, compactness %d' % (t_gen, n_seg, compact), flush=True)
    t0 = time.time()
    raw = slic(vol, n_segments=n_seg, compactness=compact, spacing=space, sigma=1, multichannel=False)
    t_slic = time.time() - t0
    del vol
    raw_crc = crc(np.asarray(raw).astype(np.int32))
    print('slic %.0f s, raw crc %08x' % (t_slic, raw_crc), flush=True)
    t0 = time.time()
    lab = measure.label(raw)
    t_label = time.time() - t0
    del raw
    lab32 = np.asarray(lab).astype(np.int32)
    nb = int(lab32.max()) + 1
    out = {'versions': np.array('scikit-image %s, numpy %s' % (skimage.__version__, np.__version__)),
           'shape': np.array(C5_FULL_SHAPE), 'volume_crc': np.array(vol_crc, dtype=np.uint32),
           'slic_raw_crc': np.array(raw_crc, dtype=np.uint32), 'slic_crc': np.array(crc(lab32), dtype=np.uint32),
           'nb_supervoxels': np.array(nb), 'seconds_one_core': np.array([t_slic, t_label])}
    np.savez_compressed(os.path.join(HERE, 'reference_c5_full.npz'), **out)
    print('config 5 full: K = %d, slic %.0f s + label %.0f s (one core)' % (nb, t_slic, t_label))


if __name__ == '__main__':
    {'c3': config3, 'c4': config4, 'c5mid': config5_mid, 'c5full': config5_full}[sys.argv[1]]()
