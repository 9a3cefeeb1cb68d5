#!/usr/bin/env python
"""Golden input/output vectors from the reference's OWN code (run in the build container only).

`import imsegm` fails here (scikit-image, gco, nibabel are absent), so the functions that do not need
those packages are lifted out of the reference's source files with `ast` and executed unchanged in a
namespace holding numpy / scipy / scikit-learn; `features_cython.pyx` is the reference's file compiled
verbatim (oracle/_ref, see oracle/build_ref.py).  Nothing of the reference is copied into the repo: only
the resulting numbers are stored, as tests/golden/*.npz.

    python tests/golden/make_golden.py            # needs /root/reference and oracle/_ref
"""
import ast
import logging
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = '/root/reference/imsegm'
sys.path.insert(0, ROOT)


def lift(path, names, namespace):
    """exec the named top-level functions / assignments of a reference file inside `namespace`"""
    tree = ast.parse(open(path).read())
    picked = []
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name in names:
            picked.append(node)
        elif isinstance(node, ast.Assign) and any(isinstance(t, ast.Name) and t.id in names for t in node.targets):
            picked.append(node)
    missing = set(names) - {getattr(n, 'name', None) or n.targets[0].id for n in picked}
    assert not missing, missing
    exec(compile(ast.Module(body=picked, type_ignores=[]), path, 'exec'), namespace)
    return namespace


def main():
    from scipy import ndimage
    from sklearn import metrics, preprocessing
    from oracle import oracle as orc
    ref_cy = orc.ref_features_cython()
    assert ref_cy is not None, 'build oracle/_ref first (python oracle/build_ref.py)'
    rng = np.random.default_rng(20240925)

    # ---- superpixels.py: graph + 3-D centres ------------------------------------------------------
    ns = {'np': np, 'logging': logging}
    lift(os.path.join(REF, 'superpixels.py'),
         ['make_graph_segment_connect_edges', 'get_segment_diffs_2d_conn4', 'get_segment_diffs_3d_conn6',
          'make_graph_segm_connect_grid2d_conn4', 'make_graph_segm_connect_grid3d_conn6', 'superpixel_centers'], ns)
    out = {}
    yy, xx = np.mgrid[:37, :53]
    seg2d = ((yy // 6) * 9 + (xx // 6) + (rng.random((37, 53)) < 0.08) * 3).astype(np.int64)     # ragged, with holes in the label set
    v, e = ns['make_graph_segm_connect_grid2d_conn4'](seg2d)
    out.update(seg2d=seg2d, seg2d_vertices=np.asarray(v), seg2d_edges=np.asarray(e))
    zz, yy, xx = np.mgrid[:7, :19, :23]
    seg3d = ((zz // 3) * 20 + (yy // 5) * 5 + (xx // 5) + (rng.random((7, 19, 23)) < 0.05) * 2).astype(np.int64)
    v, e = ns['make_graph_segm_connect_grid3d_conn6'](seg3d)
    centres = np.array([c if len(np.shape(c)) else [-1, -1, -1] for c in ns['superpixel_centers'](seg3d)], dtype=np.float64)
    out.update(seg3d=seg3d, seg3d_vertices=np.asarray(v), seg3d_edges=np.asarray(e), seg3d_centres=centres)
    np.savez_compressed(os.path.join(HERE, 'graph.npz'), **out)

    # ---- features_cython.pyx (compiled verbatim): colour 2-D and gray 3-D statistics ---------------
    out = {}
    img = (rng.random((41, 57, 3)) * 255).astype(np.float32)
    seg = ((np.mgrid[:41, :57][0] // 7) * 9 + np.mgrid[:41, :57][1] // 7).astype(np.int32)
    seg[seg == 5] = 40                                            # an unused label in between
    mean = np.asarray(ref_cy.computeColorImage2dMean(img, seg))
    energy = np.asarray(ref_cy.computeColorImage2dEnergy(img, seg))
    var = np.asarray(ref_cy.computeColorImage2dVariance(img, seg, np.array(mean, dtype=np.float32)))
    out.update(img2d=img, seg2d=seg, mean2d=mean, energy2d=energy, var2d=var)
    vol = rng.standard_normal((6, 17, 21)).astype(np.float32)
    segv = ((np.mgrid[:6, :17, :21][0] // 3) * 12 + (np.mgrid[:6, :17, :21][1] // 6) * 4 + np.mgrid[:6, :17, :21][2] // 6).astype(np.int32)
    meanv = np.asarray(ref_cy.computeGrayImage3dMean(vol, segv))
    energyv = np.asarray(ref_cy.computeGrayImage3dEnergy(vol, segv))
    varv = np.asarray(ref_cy.computeGrayImage3dVariance(vol, segv, np.array(meanv, dtype=np.float32)))
    out.update(vol=vol, segv=segv, meanv=meanv, energyv=energyv, varv=varv)
    np.savez_compressed(os.path.join(HERE, 'descriptors.npz'), **out)

    # ---- graph_cuts.py: edge weights, unary and pairwise terms -------------------------------------
    ns = {'np': np, 'logging': logging, 'metrics': metrics, 'preprocessing': preprocessing}
    lift(os.path.join(REF, 'graph_cuts.py'),
         ['MIN_UNARY_PROB', 'MAX_PAIRWISE_COST', 'MIN_MAX_EDGE_WEIGHT', 'compute_spatial_dist', 'compute_edge_model',
          'create_pairwise_matrix_uniform', 'create_pairwise_matrix_specif', 'create_pairwise_matrix', 'compute_unary_cost',
          'compute_pairwise_cost'], ns)
    K, C = 60, 3
    edges = np.unique(np.sort(rng.integers(0, K, (150, 2)), axis=1), axis=0)
    edges = edges[edges[:, 0] != edges[:, 1]].astype(np.int32)
    proba = rng.random((K, C))
    proba /= proba.sum(axis=1, keepdims=True)
    centres = [tuple(c) for c in rng.random((K, 2)) * 100]
    out = dict(edges=edges, proba=proba, centres=np.array(centres))
    for metric in ('l1', 'l2', 'lT'):
        out['edge_model_' + metric] = np.asarray(ns['compute_edge_model'](edges, proba, metric))
    out['spatial'] = np.asarray(ns['compute_spatial_dist'](centres, edges, relative=False))
    out['spatial_rel'] = np.asarray(ns['compute_spatial_dist'](centres, edges, relative=True))
    out['unary'] = np.asarray(ns['compute_unary_cost'](proba))
    out['pairwise_scalar'] = np.asarray(ns['compute_pairwise_cost'](2.0, proba.shape))
    out['pairwise_pairs'] = np.asarray(ns['compute_pairwise_cost']([((0, 1), 0.5), ((1, 2), 3.0)], proba.shape))
    gc_mat = np.array([[0.0, 1.0, 4.0], [1.0, 0.0, 2.5], [4.0, 2.5, 0.0]])
    out['pairwise_matrix_in'] = gc_mat
    out['pairwise_matrix'] = np.asarray(ns['compute_pairwise_cost'](gc_mat, proba.shape))
    np.savez_compressed(os.path.join(HERE, 'graph_cut_terms.npz'), **out)

    # ---- descriptors.py: Leung-Malik bank and numpy statistics -------------------------------------
    from scipy.ndimage import gaussian_filter, gaussian_filter1d, gaussian_laplace
    ns = {'np': np, 'logging': logging, 'ndimage': ndimage, 'gaussian_filter': gaussian_filter,
          'gaussian_filter1d': gaussian_filter1d, 'gaussian_laplace': gaussian_laplace}
    lift(os.path.join(REF, 'descriptors.py'),
         ['DEFAULT_FILTERS_SIGMAS', 'SHORT_FILTERS_SIGMAS', 'make_gaussian_filter1d', 'make_edge_filter2d',
          'create_filter_bank_lm_2d', 'compute_img_filter_response2d', 'image_subtract_gauss_smooth'], ns)
    bank, names = ns['create_filter_bank_lm_2d'](radius=8, sigmas=(np.sqrt(2), 2), nb_orient=4)
    gray = rng.random((40, 48))
    responses = [np.asarray(ns['compute_img_filter_response2d'](gray, battery)) for battery in bank]
    smooth = np.asarray(ns['image_subtract_gauss_smooth'](rng.random((30, 36, 3)) * 255, 3.))
    np.savez_compressed(os.path.join(HERE, 'texture.npz'), gray=gray, names=np.array(names),
                        **{'battery_%02d' % i: np.asarray(b) for i, b in enumerate(bank)},
                        **{'response_%02d' % i: r for i, r in enumerate(responses)}, smooth=smooth)

    # ---- labeling.py: superpixel x annotation histograms (supervised path, pipelines.py:284) ---------
    class ImageDimensionError(TypeError):
        pass
    ns = {'np': np, 'logging': logging, 'ImageDimensionError': ImageDimensionError}
    lift(os.path.join(REF, 'labeling.py'), ['histogram_regions_labels_counts', 'histogram_regions_labels_norm'], ns)
    rng2 = np.random.default_rng(77)                       # own stream: the sections above keep their vectors
    yy, xx = np.mgrid[:45, :61]
    slic = ((yy // 7) * 10 + xx // 7).astype(np.int64)
    slic[slic == 13] = 80                                  # an unused stretch of labels
    annot = ((yy > 20).astype(np.int64) + (xx > 33) * 2)
    annot[rng2.random(annot.shape) < 0.15] = 5            # sparse extra label, leaves label 4 unused
    # driver touch point run_segm_slic_model_graphcut.py:373,422: background label on the image boundary
    lift(os.path.join(REF, 'utilities', 'data_io.py'), ['get_image2d_boundary_color'], ns)
    lift(os.path.join(REF, 'labeling.py'), ['assume_bg_on_boundary'], ns)
    segm_a = (annot % 3).astype(np.int64)                  # boundary dominated by label 0 or 1, all labels in use
    segm_b = segm_a + 1                                    # background label 0 not in use
    segm_b[10:30, 10:40] = 3
    bg = {}
    for name, sg in (('a', segm_a), ('b', segm_b)):
        for size in (1, 3):
            bg['bg_%s_%d' % (name, size)] = ns['assume_bg_on_boundary'](sg.copy(), bg_label=0, boundary_size=size)
    colour = rng2.integers(0, 255, (9, 14, 3))
    np.savez_compressed(os.path.join(HERE, 'labeling.npz'), slic=slic, annot=annot,
                        counts=ns['histogram_regions_labels_counts'](slic, annot),
                        norm=ns['histogram_regions_labels_norm'](slic, annot),
                        segm_a=segm_a, segm_b=segm_b, colour=colour,
                        colour_bg=ns['get_image2d_boundary_color'](colour, size=2), **bg)
    print('golden vectors written to', HERE)


if __name__ == '__main__':
    main()
