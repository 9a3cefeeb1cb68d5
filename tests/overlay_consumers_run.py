#!/usr/bin/env python
"""The OTHER consumers of the kernels among the reference's unchanged files, run on top of this repo's
`imsegm` overlay package (test infrastructure, the sibling of tests/overlay_driver_run.py; started by
tests/test_overlay_consumers.py / tests/test_gpu_zz_consumers.py under the image's conda Python 3.9):

    /opt/conda/bin/python3.9 tests/overlay_consumers_run.py <reference tree> <out_dir> [--device]

1. `experiments_segmentation/run_eval_superpixels.py` -- `compute_boundary_distance` with `--slico`
   (/root/reference/experiments_segmentation/run_eval_superpixels.py:108-131): SLIC and SLICO of an ovary slice against its egg
   annotation, the mean boundary distance of each;
2. `experiments_segmentation/run_segm_slic_classif_graphcut.py` -- `load_image_annot_compute_features_labels` for two annotated
   images, the classifier the driver trains (`seg_clf.create_classif_search_train_export`) and `segment_image` with it
   (/root/reference/experiments_segmentation/run_segm_slic_classif_graphcut.py:184-228, 323-385);
3. `imsegm/ellipse_fitting.py` -- `get_slic_points_labels` and `prepare_boundary_points_close`
   (/root/reference/imsegm/ellipse_fitting.py:264-279, 625-645; the latter against the points its own doctest prints).

Nothing of those files is edited or copied: the modules are imported from the reference tree (or from the bundle oracle/build_ref.py
stages for the GPU box), and their `imsegm.superpixels / descriptors / labeling / pipelines / graph_cuts / classification` are this
repo's modules.  Without `--device` the ctypes session classes are replaced by the oracle-backed stand-ins of tests/dryrun_plugin.py
(import graph and glue only).  Prints one JSON line `CONSUMERS {...}`.
"""
import json
import os
import sys
import types
import warnings

warnings.filterwarnings('ignore')
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def main():
    ref, out_dir = os.path.abspath(sys.argv[1]), os.path.abspath(sys.argv[2])
    use_device = '--device' in sys.argv[3:]
    sys.path[:0] = [ROOT, ref, os.path.join(ref, 'experiments_segmentation')]
    for name in ('nibabel', 'planar', 'OleFileIO_PL'):
        if name not in sys.modules:
            try:
                __import__(name)
            except ImportError:
                sys.modules[name] = types.ModuleType(name)
    if not hasattr(sys.modules['planar'], 'line'):
        sys.modules['planar'].line = types.ModuleType('planar.line')
    os.chdir(ref)                                   # the scripts resolve 'data-images' relative to the tree
    if not use_device:
        sys.path.insert(0, HERE)
        import dryrun_plugin
        dryrun_plugin.pytest_configure(None)
    import numpy as np
    for alias, typ in (('float', float), ('int', int), ('bool', bool)):     # the reference predates numpy 1.24
        if alias not in np.__dict__:
            setattr(np, alias, typ)
    import imsegm
    assert imsegm.REFERENCE_PATH == os.path.join(ref, 'imsegm'), imsegm.REFERENCE_PATH
    from pyimsegm_amd import _hip
    calls = {'slic': 0, 'slico': 0, 'label_hist': 0, 'graph': 0, 'segment_or_cut': 0}
    if use_device:                                  # count what reaches libimsegm_hip.so
        real_slic, real_hist, real_graph = _hip.Image2D.slic, _hip.Image2D.label_hist, _hip.Image2D.graph
        real_segment, real_cut = _hip.Image2D.segment, _hip.cut_general_graph

        def counted_slic(self, *a, **kw):
            calls['slico' if kw.get('slic_zero') else 'slic'] += 1
            return real_slic(self, *a, **kw)

        def counted_hist(self, *a, **kw):
            calls['label_hist'] += 1
            return real_hist(self, *a, **kw)

        def counted_graph(self, *a, **kw):
            calls['graph'] += 1
            return real_graph(self, *a, **kw)

        def counted_segment(self, *a, **kw):
            calls['segment_or_cut'] += 1
            return real_segment(self, *a, **kw)

        def counted_cut(*a, **kw):
            calls['segment_or_cut'] += 1
            return real_cut(*a, **kw)
        _hip.Image2D.slic, _hip.Image2D.label_hist, _hip.Image2D.graph = counted_slic, counted_hist, counted_graph
        _hip.Image2D.segment, _hip.cut_general_graph = counted_segment, counted_cut
    import pyimsegm_amd.superpixels
    import pyimsegm_amd.pipelines
    import pyimsegm_amd.labeling
    seen = {}
    data = os.path.join(ref, 'data-images', 'drosophila_ovary_slice')

    # ---- 1. run_eval_superpixels.py: SLIC and SLICO against the egg annotation
    import run_eval_superpixels as ev               # the reference's script, unchanged
    seen['eval_superpixels_is_hip'] = ev.seg_spx is pyimsegm_amd.superpixels
    row = {'path_image': os.path.join(data, 'image', 'insitu7545.jpg'), 'path_segm': os.path.join(data, 'annot_eggs', 'insitu7545.png')}
    dists = {}
    for slico in (False, True):
        params = {'img_type': '2d_split', 'slic_size': 20, 'slic_regul': 0.25, 'slico': slico}
        name, dist = ev.compute_boundary_distance((0, row), params, path_out='')
        dists['slico' if slico else 'slic'] = float(dist)
    seen['eval_name'] = name
    seen['eval_mean_boundary_distance'] = dists

    # ---- 2. run_segm_slic_classif_graphcut.py: features + labels of annotated images, the driver's classifier, segment_image
    import run_segm_slic_classif_graphcut as sup    # the reference's script, unchanged
    seen['classif_driver_is_hip'] = bool(sup.seg_pipe is pyimsegm_amd.pipelines and sup.seg_spx is pyimsegm_amd.superpixels
                                         and sup.seg_label is pyimsegm_amd.labeling)
    params = dict(sup.SEGM_PARAMS)
    params.update({'features': sup.FEATURES_SET_COLOR, 'path_exp': out_dir, 'nb_classif_search': 2, 'gc_regul': 2.0,
                   'visual': False})
    for sub in sup.LIST_FOLDERS_BASE + sup.LIST_FOLDERS_DEBUG:
        os.makedirs(os.path.join(out_dir, sub), exist_ok=True)
    dict_features, dict_labels, shapes = {}, {}, {}
    for i, stem in enumerate(('insitu4174', 'insitu7545')):
        row = {'path_image': os.path.join(data, 'image', stem + '.jpg'), 'path_annot': os.path.join(data, 'annot_struct', stem + '.png')}
        idx_name, img, annot, slic, features, labels, label_hist, feature_names = \
            sup.load_image_annot_compute_features_labels((i, row), params, show_debug_imgs=False)
        assert features.shape[0] == slic.max() + 1 == len(labels) == label_hist.shape[0], (features.shape, slic.max())
        dict_features[idx_name], dict_labels[idx_name], shapes[idx_name] = features, labels, list(slic.shape)
    seen['train_images'] = shapes
    seen['feature_names'] = list(feature_names)
    feats, labs, sizes = sup.seg_clf.convert_set_features_labels_2_dataset(dict_features, dict_labels, balance_type='unique')
    np.random.seed(0)
    classif, path_classif = sup.seg_clf.create_classif_search_train_export(
        params['classif'], feats, labs, cross_val=sup.seg_clf.CrossValidateGroups(sizes, nb_hold_out=1),
        nb_search_iter=params['nb_classif_search'], nb_workers=1, path_out=out_dir, params=params, pca_coef=None,
        feature_names=feature_names)
    seen['classifier_files'] = sorted(f for f in os.listdir(out_dir) if 'classif' in f.lower())
    path_img = os.path.join(data, 'image', 'insitu7545.jpg')
    path_out = os.path.join(out_dir, sup.FOLDER_TRAIN)
    idx_name, segm_map, segm_gc = sup.segment_image((0, path_img), params, classif, path_out, path_visu=None, show_debug_imgs=False)
    annot = sup.load_image(os.path.join(data, 'annot_struct', 'insitu7545.png'), '2d_segm')
    seen.update(segment_name=idx_name, segment_shape=list(segm_gc.shape), segment_classes=sorted(int(v) for v in np.unique(segm_gc)),
                segment_files=sorted(os.listdir(path_out)),
                segment_agrees_with_annot=float(np.mean(segm_gc == annot)), map_agrees_with_annot=float(np.mean(segm_map == annot)))

    # ---- 3. ellipse_fitting: SLIC of a segmentation, centres, boundary points (the module's own doctest, :625-640)
    import imsegm.ellipse_fitting as ell
    seen['ellipse_fitting'] = os.path.relpath(ell.__file__, ref)
    seen['ellipse_fitting_is_hip'] = ell.segment_slic_img2d is pyimsegm_amd.superpixels.segment_slic_img2d
    seg = np.zeros((100, 200), dtype=int)
    seg = ell.add_overlap_ellipse(seg, (50, 100, 40, 60, np.deg2rad(30)), 1)
    slic, centres, labels = ell.get_slic_points_labels(seg, slic_size=15, slic_regul=0.1)
    seen['ellipse_slic'] = {'shape': list(slic.shape), 'superpixels': int(slic.max()) + 1, 'centres': list(centres.shape),
                            'labels_found': sorted(int(v) for v in np.unique(labels))}
    pts = ell.prepare_boundary_points_close(seg, [(40, 90)])
    seen['ellipse_boundary_points'] = {'count': int(len(pts[0])), 'first': pts[0][0].tolist(), 'second': pts[0][1].tolist(),
                                       'last': pts[0][-1].tolist()}
    seen['device_calls'] = calls
    if use_device:
        seen['library'] = os.path.relpath(_hip.LIB_PATH, ROOT)
    print('CONSUMERS ' + json.dumps(seen))


if __name__ == '__main__':
    main()
