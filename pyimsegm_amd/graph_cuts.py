"""GraphCut stage: class model, unary / pairwise / edge terms and the alpha-expansion cut.

Host-side mirror of the reference module ``imsegm/graph_cuts.py`` (same public names and argument
meaning).  The model fit stays on the host in scikit-learn exactly as in the reference
(``graph_cuts.py:73-163``); the region adjacency graph, the superpixel centres and the
alpha-expansion itself (``gco.cut_general_graph`` in the reference) run in the HIP library.
The small per-edge weight formulas (E ~ 5e3 values) are evaluated with the same numpy /
scikit-learn calls as the reference so that the integer energies handed to the cut are identical.
"""
import logging

import numpy as np
from sklearn import cluster, decomposition, metrics, mixture, pipeline, preprocessing
from sklearn.utils.extmath import row_norms

from pyimsegm_amd import _hip
from pyimsegm_amd.descriptors import compute_selected_features_img2d
from pyimsegm_amd.superpixels import (
    _graph_from_session,
    _session_for_labels,
    make_graph_segm_connect_grid2d_conn4,
    make_graph_segm_connect_grid3d_conn6,
    superpixel_centers,
)

#: define number of iteration in Graph-Cut optimization
DEFAULT_GC_ITERATIONS = 25
#: define minimal value of unary (being a class) term in Graph-Cut
MIN_UNARY_PROB = 0.01
#: define maximal value of pairwise (smoothness) term in Graph-Cut
MAX_PAIRWISE_COST = 1e5
#: max is this value and min is inverse (1 / val)
MIN_MAX_EDGE_WEIGHT = 1e3


def cut_grid_graph(unary_cost, pairwise_cost, cost_v, cost_h, n_iter=-1, algorithm='expansion', **kwargs):
    """ drop-in for ``gco.cut_grid_graph`` (reference ``region_growing.py:20,248``) on the GPU """
    return _hip.cut_grid_graph(unary_cost, pairwise_cost, cost_v, cost_h, n_iter=n_iter, algorithm=algorithm)


def cut_general_graph(edges, edge_weights, unary_cost, pairwise_cost, n_iter=-1, algorithm='expansion', **kwargs):
    """ drop-in for ``gco.cut_general_graph`` (reference import ``graph_cuts.py:12-15``) on the GPU """
    return _hip.cut_general_graph(edges, edge_weights, unary_cost, pairwise_cost, n_iter=n_iter, algorithm=algorithm)


def predict_proba(model, features):
    """ ``model.predict_proba(features)`` without scikit-learn's per-call input validation for the model the
    pipelines fit (``Pipeline([StandardScaler, GaussianMixture])``, :func:`estim_class_model`): the very same
    arithmetic (``StandardScaler.transform``, ``GaussianMixture._estimate_log_prob_resp``), bit for bit; any
    other model goes through its own ``predict_proba``.  Worth a third of the host time of a pipeline step
    when several images are in flight (the worker threads share the interpreter lock). """
    try:
        from scipy.special import logsumexp
        from sklearn.mixture import GaussianMixture
        from sklearn.pipeline import Pipeline
        steps = model.steps if isinstance(model, Pipeline) else None
        if steps and type(steps[-1][1]) is GaussianMixture and hasattr(steps[-1][1], '_estimate_weighted_log_prob') \
                and all(type(st) is preprocessing.StandardScaler for _, st in steps[:-1]):
            feats = np.array(features, dtype=np.float64)
            if feats.ndim != 2 or not np.isfinite(feats).all():
                return model.predict_proba(features)
            for _, scaler in steps[:-1]:
                if scaler.with_mean:
                    feats -= scaler.mean_
                if scaler.with_std:
                    feats /= scaler.scale_
            weighted = steps[-1][1]._estimate_weighted_log_prob(feats)
            log_prob_norm = logsumexp(weighted, axis=1)
            with np.errstate(under='ignore'):
                log_resp = weighted - log_prob_norm[:, np.newaxis]
            return np.exp(log_resp)
    except Exception:       # private scikit-learn API moved: fall back to the public call
        pass
    return model.predict_proba(features)


def estim_gmm_params(features, prob):
    """ GMM parameters from a soft labelling (arg-max assignment)

    >>> np.random.seed(0)
    >>> prob = np.array([[1, 0]] * 30 + [[0, 1]] * 40)
    >>> fts = prob + np.random.random(prob.shape)
    >>> mm = estim_gmm_params(fts, prob)
    >>> mm['weights']
    [0.42857142857142855, 0.5714285714285714]
    >>> np.round(mm['means'], 4).tolist()
    [[1.4954, 0.5375], [0.542, 1.4261]]
    """
    nb_samples, nb_classes = prob.shape
    labels = np.argmax(prob, axis=1)
    params = {'weights': [], 'means': [], 'covars': []}
    for lb in range(nb_classes):
        sel = labels == lb
        params['weights'].append(float(np.sum(sel)) / float(nb_samples))
        params['means'].append(np.mean(features[sel], axis=0))
        params['covars'].append(np.cov(features[sel]))
    params['means'] = np.array([m.tolist() for m in params['means']])
    # np.cov(samples) as the reference calls it yields one (n_c x n_c) matrix per class: ragged
    return params


def threshold_otsu(values, nbins=256):
    """ Otsu threshold of a 1D sample (numpy restatement of ``skimage.filters.threshold_otsu``) """
    values = np.asarray(values, dtype=np.float64).ravel()
    hist, edges = np.histogram(values, bins=nbins, range=(values.min(), values.max()))
    centers = (edges[:-1] + edges[1:]) / 2.
    hist = hist.astype(float)
    w1 = np.cumsum(hist)
    w2 = np.cumsum(hist[::-1])[::-1]
    with np.errstate(invalid='ignore', divide='ignore'):
        m1 = np.cumsum(hist * centers) / w1
        m2 = (np.cumsum((hist * centers)[::-1]) / w2[::-1])[::-1]
    var12 = w1[:-1] * w2[1:] * (m1[:-1] - m2[1:])**2
    return centers[:-1][np.nanargmax(var12)]


def compute_multivarian_otsu(features):
    """ Otsu per feature dimension with majority vote on orientation

    >>> np.random.seed(0)
    >>> fts = np.vstack([np.random.random((5, 3)) - 1, np.random.random((5, 3)) + 1])
    >>> fts[:, 1] = - fts[:, 1]
    >>> compute_multivarian_otsu(fts).astype(int)
    array([0, 0, 0, 0, 0, 1, 1, 1, 1, 1])
    """
    ys = np.zeros(features.shape)
    for i in range(features.shape[-1]):
        asign = features[:, i] > threshold_otsu(features[:, i])
        if i > 0:
            m = np.mean(ys[:, :i], axis=1)
            if np.mean(np.abs(~asign - m)) < np.mean(np.abs(asign - m)):
                asign = ~asign
        ys[:, i] = asign
    return np.mean(ys, axis=1) > 0.5


def estim_class_model(features, nb_classes, estim_model='GMM', pca_coef=None, use_scaler=True, max_iter=99):
    """ scikit-learn pipeline (scaler, PCA, mixture model) fitted on the superpixel features;
    identical construction to the reference (``graph_cuts.py:73-163``), host side on purpose

    >>> np.random.seed(0)
    >>> fts = np.vstack([np.random.random((50, 3)) - 1, np.random.random((50, 3)) + 1])
    >>> mm = estim_class_model(fts, 2)
    >>> mm.predict_proba(fts).shape
    (100, 2)
    >>> mm = estim_class_model(fts, 2, estim_model='GMM_kmeans', pca_coef=0.95, max_iter=3)
    >>> mm.predict_proba(fts).shape
    (100, 2)
    >>> mm = estim_class_model(fts, 2, estim_model='GMM_Otsu', max_iter=3)
    >>> mm.predict_proba(fts).shape
    (100, 2)
    >>> mm = estim_class_model(fts, 2, estim_model='kmeans_quantiles', use_scaler=False, max_iter=3)
    >>> mm.predict_proba(fts).shape
    (100, 2)
    >>> mm = estim_class_model(fts, 2, estim_model='BGM', max_iter=3)
    >>> mm.predict_proba(fts).shape
    (100, 2)
    >>> mm = estim_class_model(fts, 2, estim_model='Otsu', max_iter=3)
    >>> mm.predict_proba(fts).shape
    (100, 2)
    """
    components = []
    if use_scaler:
        components.append(('std_scaler', preprocessing.StandardScaler()))
    if pca_coef is not None:
        components.append(('reduce_dim', decomposition.PCA(pca_coef)))
    nb_inits = max(1, int(np.sqrt(max_iter)))
    mm = mixture.GaussianMixture(n_components=nb_classes, covariance_type='full', n_init=nb_inits, max_iter=max_iter)
    if '_' in estim_model:
        estim_model, init_type = estim_model.split('_')[0], estim_model.split('_')[-1]
    else:
        init_type = ''
    y = None
    if estim_model == 'GMM':
        if init_type == 'kmeans':
            mm.set_params(n_init=1)
            y = cluster.KMeans(n_clusters=nb_classes, init='k-means++').fit_predict(features)
        elif init_type == 'Otsu':
            mm.set_params(n_init=1)
            y = compute_multivarian_otsu(features)
    elif estim_model == 'kmeans':
        mm.set_params(max_iter=1)
        init_type = 'quantiles' if init_type == 'quantiles' else 'k-means++'
        _, y = estim_class_model_kmeans(features, nb_classes, init_type=init_type, max_iter=max_iter)
        logging.info('compute probability of each feature to all component')
    elif estim_model == 'BGM':
        mm = mixture.BayesianGaussianMixture(n_components=nb_classes, covariance_type='full', n_init=nb_inits,
                                             max_iter=max_iter)
    elif estim_model == 'Otsu' and nb_classes == 2:
        mm.set_params(max_iter=1, n_init=1)
        y = compute_multivarian_otsu(features)
    components.append(('model', mm))
    model = pipeline.Pipeline(components)
    if y is not None:
        model.fit(features, y)
    else:
        model.fit(features)
    return model


def estim_class_model_gmm(features, nb_classes, init='kmeans'):
    """ Gaussian mixture over the features, optionally after a k-means pass

    >>> np.random.seed(0)
    >>> fts = np.vstack([np.random.random((50, 3)) - 1, np.random.random((50, 3)) + 1])
    >>> estim_class_model_gmm(fts, 2).predict_proba(fts).shape
    (100, 2)
    """
    logging.debug('estimate GMM for all given features %r and %i component', features.shape, nb_classes)
    gmm = mixture.GaussianMixture(n_components=nb_classes, covariance_type='full', max_iter=99)
    if init == 'kmeans':
        y = cluster.KMeans(n_clusters=nb_classes, init='k-means++').fit_predict(features)
        gmm.fit(features, y)
    else:
        gmm.fit(features)
    return gmm


def estim_class_model_kmeans(features, nb_classes, init_type='k-means++', max_iter=99):
    """ k-means clustering followed by a one-step Gaussian mixture

    >>> np.random.seed(0)
    >>> fts = np.vstack([np.random.random((50, 3)) - 1, np.random.random((50, 3)) + 1])
    >>> mm, y = estim_class_model_kmeans(fts, 2, max_iter=9)
    >>> y.shape
    (100,)
    >>> mm.predict_proba(fts).shape
    (100, 2)
    """
    if init_type == 'quantiles':
        quantiles = np.linspace(5, 95, nb_classes).tolist()
        init_perc = np.array(np.percentile(features, quantiles, axis=0))
        kmeans = cluster.KMeans(nb_classes, init=init_perc, max_iter=2, n_init=1)
    else:
        nb_inits = max(1, int(np.sqrt(max_iter)))
        kmeans = cluster.KMeans(nb_classes, init=init_type, max_iter=max_iter, n_init=nb_inits)
    y = kmeans.fit_predict(features)
    gmm = mixture.GaussianMixture(n_components=nb_classes, covariance_type='full', max_iter=1)
    gmm.fit(features, y)
    return gmm, y


def get_vertexes_edges(segments):
    """ (vertices, edges) of the region adjacency graph of a 2D / 3D label map """
    segments = np.asarray(segments)
    if segments.ndim == 3:
        return make_graph_segm_connect_grid3d_conn6(segments)
    if segments.ndim == 2:
        return make_graph_segm_connect_grid2d_conn4(segments)
    return None, None


def compute_spatial_dist(centres, edges, relative=False):
    """ Euclidean distance between the centres of connected superpixels

    >>> centres = [(0.5, 1.0), (0.0, 3.5), (0.0, 7.0), [-1, -1], (1.0, 1.5), (1.0, 4.5), (1.0, 8.0)]
    >>> edges = [[0, 1], [1, 2], [4, 5], [5, 6], [0, 4], [1, 5], [2, 6]]
    >>> np.round(compute_spatial_dist(centres, edges), 2).tolist()
    [2.55, 3.5, 3.0, 3.5, 0.71, 1.41, 1.41]
    """
    if np.max(edges) >= len(centres):
        raise ValueError('max vertex %i exceed size of centres %i' % (np.max(edges), len(centres)))
    if isinstance(centres, np.ndarray) and centres.ndim == 2 and centres.dtype == np.float64:
        centres = np.nan_to_num(centres)      # dense table straight from the device
    else:
        centres = list(centres)
        ndim = np.max([len(c) for c in centres if c is not None])
        for i, c in enumerate(centres):
            if c is None or len(c) == 0:
                centres[i] = [np.nan] * ndim
        centres = np.nan_to_num(np.asarray(centres, dtype=np.float64))
    edges = np.asarray(edges)
    # == sklearn.metrics.pairwise.paired_euclidean_distances (bit for bit) without its input validation
    dist = row_norms(centres[edges[:, 0]] - centres[edges[:, 1]])
    if relative:
        dist = dist / np.mean(dist)
    return dist


def compute_edge_model(edges, proba, metric='l_T'):
    """ edge weights from the class probabilities of the two end superpixels:
    ``exp(-dist / (2 * std(dist)**2))`` with an l1, l2 or max-squared-difference distance

    >>> edges = np.array([[0, 1], [1, 2], [0, 4], [1, 4], [1, 5], [2, 5], [4, 5], [2, 6], [5, 6]])
    >>> np.random.seed(0)
    >>> img = np.random.random((2, 12, 3)) * 255
    >>> proba = np.random.random((7, 2))
    >>> np.round(compute_edge_model(edges, proba, metric='l1'), 3).tolist()
    [0.002, 0.015, 0.001, 0.002, 0.0, 0.002, 0.015, 0.034, 0.001]
    >>> np.round(compute_edge_model(edges, proba, metric='lT'), 3).tolist()
    [0.0, 0.002, 0.0, 0.005, 0.0, 0.0, 0.101, 0.092, 0.001]
    """
    edges = np.asarray(edges)
    if np.max(edges) >= len(proba):
        raise ValueError('max vertex %i exceed size of proba %r' % (np.max(edges), proba.shape))
    v1, v2 = proba[edges[:, 0]], proba[edges[:, 1]]
    if metric == 'l1':
        dist = metrics.pairwise.paired_manhattan_distances(v1, v2)
    elif metric == 'l2':
        dist = metrics.pairwise.paired_euclidean_distances(v1, v2)
    elif metric == 'lT':
        dist = np.max((v1 - v2)**2, axis=1)
    else:
        logging.error('not implemented for: %s', metric)
        return np.ones(len(edges))
    return np.exp(-dist / (2 * np.std(dist)**2))


def create_pairwise_matrix_uniform(gc_reg, nb_classes):
    """ uniform pairwise matrix with zero diagonal

    >>> create_pairwise_matrix_uniform(0.2, 3).tolist()
    [[0.0, 0.2, 0.2], [0.2, 0.0, 0.2], [0.2, 0.2, 0.0]]
    """
    return (np.ones(nb_classes) - np.eye(nb_classes)) * gc_reg


def create_pairwise_matrix_specif(pos_weights, nb_classes=None):
    """ pairwise matrix of ones with specific symmetric entries

    >>> create_pairwise_matrix_specif([((1, 2), 0.5), ((1, 0), 0.7)], 4).tolist()
    [[0.0, 0.7, 1.0, 1.0], [0.7, 0.0, 0.5, 1.0], [1.0, 0.5, 0.0, 1.0], [1.0, 1.0, 1.0, 0.0]]
    """
    if not nb_classes:
        nb_classes = np.max([list(c) for c, _ in pos_weights]) + 1
    pairwise = np.ones(nb_classes) - np.eye(nb_classes)
    for (i, j), w in pos_weights:
        pairwise[i, j] = pairwise[j, i] = w
    return pairwise


def create_pairwise_matrix(gc_regul, nb_classes):
    """ pairwise matrix from a scalar, a list of specific entries or a full matrix

    >>> create_pairwise_matrix(0.6, 3).tolist()
    [[0.0, 0.6, 0.6], [0.6, 0.0, 0.6], [0.6, 0.6, 0.0]]
    >>> create_pairwise_matrix([((1, 2), 0.5), ((0, 2), 0.7)], 3).tolist()
    [[0.0, 1.0, 0.7], [1.0, 0.0, 0.5], [0.7, 0.5, 0.0]]
    """
    if isinstance(gc_regul, np.ndarray):
        if not gc_regul.shape[0] == gc_regul.shape[1] == nb_classes:
            raise ValueError('GC regul matrix %r should match match number of classes (%i)' %
                             (gc_regul.shape, nb_classes))
        return gc_regul - np.min(gc_regul)
    if isinstance(gc_regul, list):
        return create_pairwise_matrix_specif(gc_regul, nb_classes)
    return create_pairwise_matrix_uniform(gc_regul, nb_classes)


def compute_unary_cost(proba, min_prob=MIN_UNARY_PROB):
    """ ``|-log(clip(proba, min_prob, 1 - min_prob))|``

    >>> compute_unary_cost(np.array([[0.5, 0.001], [1., 0.3]])).round(4).tolist()
    [[0.6931, 4.6052], [0.0101, 1.204]]
    """
    proba = np.array(proba, dtype=np.float64)
    proba[proba < min_prob] = min_prob
    proba[proba > 1 - min_prob] = 1 - min_prob
    return np.abs(np.array(-np.log(proba), dtype=np.float64))


def compute_pairwise_cost(gc_regul, proba_shape, max_pairwise_cost=MAX_PAIRWISE_COST):
    """ pairwise cost matrix clipped at ``max_pairwise_cost`` """
    pairwise_cost = np.array(create_pairwise_matrix(gc_regul, proba_shape[1]), dtype=np.float64)
    pairwise_cost[pairwise_cost > max_pairwise_cost] = max_pairwise_cost
    return pairwise_cost


def _reference_drawing():
    """``imsegm.utilities.drawing`` of an installed reference (overlay mode of the ``imsegm`` package), else None"""
    import sys
    pkg = sys.modules.get('imsegm')
    if pkg is None or getattr(pkg, 'REFERENCE_PATH', None) is None:
        return None
    try:
        import importlib
        return importlib.import_module('imsegm.utilities.drawing')
    except Exception as ex:        # drawing needs matplotlib / planar: not part of the hot path
        logging.debug('reference drawing module not importable: %s', ex)
        return None


def insert_gc_debug_images(debug_visual, segments, graph_labels, unary_cost, edges, edge_weights):
    """ store intermediate variables; when the reference's drawing module is installed (``imsegm`` overlay) also
    the rendered debug images of ``graph_cuts.py:558-571`` (``imgs_unary_cost``, ``img_graph_edges``,
    ``img_graph_segm``) that ``drawing.figure_segm_graphcut_debug`` expects """
    if debug_visual is None:
        return
    debug_visual['segments'] = segments
    debug_visual['edges'] = edges
    debug_visual['edge_weights'] = edge_weights
    debug_visual['unary_cost'] = unary_cost
    debug_visual['graph_labels'] = graph_labels
    draw = _reference_drawing()
    if draw is None:
        return
    segments = np.asarray(segments)
    debug_visual['imgs_unary_cost'] = draw.draw_graphcut_unary_cost_segments(segments, unary_cost)
    from pyimsegm_amd.superpixels import superpixel_centers
    debug_visual['img_graph_edges'] = draw.draw_graphcut_weighted_edges(
        segments, superpixel_centers(segments), edges, edge_weights, img_bg=debug_visual.get('slic_mean', None))
    debug_visual['img_graph_segm'] = draw.draw_color_labeling(segments, graph_labels)


def _edges_centres(segments, _session=None):
    own = _session is None
    sess = _session_for_labels(segments) if own else _session
    _, edges, centres, present = _graph_from_session(sess)
    if own:
        sess.close()
    return edges, centres, present


def compute_edge_weights(segments, image=None, features=None, proba=None, edge_type='', _session=None):
    """ edges of the superpixel graph and their weights (reference ``graph_cuts.py:574-657``)

    :param ndarray segments: superpixels
    :param ndarray image: input image (``edge_type='color'``)
    :param ndarray features: superpixel features (``edge_type='features'``)
    :param ndarray proba: class probabilities (``edge_type='model[_l1|_l2|_lT]'``)
    :param str edge_type: '', 'const', 'spatial', 'color', 'features', 'model', 'model_<metric>'
    :return tuple(ndarray,ndarray): int32 edges E x 2, float weights E clipped to [1e-3, 1e3]
    """
    logging.debug('extraction segment connectivity...')
    if _session is None:
        segments = np.asarray(segments)
    if segments.ndim in (2, 3):
        edges, centres, present = _edges_centres(segments, _session)
        edges = np.array(edges, dtype=np.int32).reshape(-1, 2)
        centre_list = None
    else:
        _, edges = get_vertexes_edges(segments)
        edges = np.array(edges, dtype=np.int32)
        centres = present = None
        centre_list = superpixel_centers(segments)
    logging.debug('graph edges %r', edges.shape)

    edge_weights = edge_weights_from_graph(edges, centres if centre_list is None else centre_list, features, proba,
                                           edge_type, image=image, segments=segments)
    return edges, edge_weights


def edge_weights_from_graph(edges, centres, features=None, proba=None, edge_type='', image=None, segments=None):
    """ the weights of given edges (second half of :func:`compute_edge_weights`, reference ``graph_cuts.py:616-657``)

    :param ndarray edges: int32 E x 2
    :param centres: superpixel centres, dense K x ndim table or list of tuples
    :return ndarray: float weights E clipped to [1e-3, 1e3]
    """
    if edge_type.startswith('model'):
        if proba is None or len(proba) == 0:
            raise ValueError('"proba" is required')
        metric = edge_type.split('_')[-1] if '_' in edge_type else 'lT'
        edge_weights = compute_edge_model(edges, proba, metric)
    elif edge_type == 'color':
        if image is None:
            raise RuntimeError('"image" is required')
        image_float = np.array(image, dtype=float)
        if np.max(image) > 1:
            image_float /= 255.
        color, _ = compute_selected_features_img2d(image_float, segments, {'color': ['mean']})
        dist = metrics.pairwise.paired_manhattan_distances(color[edges[:, 0]], color[edges[:, 1]])
        edge_weights = np.exp(-(dist.astype(float) / (2 * np.std(dist)**2)))
    elif edge_type == 'features':
        if features is None:
            raise RuntimeError('"features" is required')
        features_norm = preprocessing.StandardScaler().fit_transform(features)
        dist = metrics.pairwise.paired_euclidean_distances(features_norm[edges[:, 0]], features_norm[edges[:, 1]])
        edge_weights = np.exp(-(dist.astype(float) / (2 * np.std(dist)**2)))
    else:
        edge_weights = np.ones(len(edges))

    edge_weights = np.array(edge_weights, dtype=float)
    if edge_type in ['model', 'features', 'color', 'spatial']:
        # device centres already hold [-1, -1] for unused labels (superpixels.py:218 semantics)
        edge_weights /= compute_spatial_dist(centres, edges, relative=True)

    edge_weights[edge_weights < 1. / MIN_MAX_EDGE_WEIGHT] = 1. / MIN_MAX_EDGE_WEIGHT
    edge_weights[edge_weights > MIN_MAX_EDGE_WEIGHT] = MIN_MAX_EDGE_WEIGHT
    return edge_weights


def segment_graph_cut_general(
    segments,
    proba,
    image=None,
    features=None,
    gc_regul=1.,
    edge_type='model',
    edge_cost=1.,
    debug_visual=None,
    _session=None,
):
    """ label the superpixels by an alpha-expansion graph cut (reference ``graph_cuts.py:660-747``)

    :param ndarray segments: superpixel label map
    :param ndarray proba: probabilities K x C of each superpixel belonging to each class
    :param ndarray image: image (only for ``edge_type='color'``)
    :param ndarray features: features (only for ``edge_type='features'``)
    :param gc_regul: regularisation: scalar, list of specific pairs or full matrix
    :param str edge_type: see :func:`compute_edge_weights`
    :param float edge_cost: global multiplier of the edge weights
    :param dict debug_visual: filled with intermediate results if given
    :return ndarray: int32 class per superpixel
    """
    logging.debug('convert variables and run GraphCut on created graph.')
    proba = np.asarray(proba, dtype=np.float64)
    edges, edge_weights = compute_edge_weights(segments, image, features, proba, edge_type, _session=_session)
    edge_weights *= edge_cost
    unary_cost = compute_unary_cost(proba)
    pairwise_cost = compute_pairwise_cost(gc_regul, proba.shape)
    logging.debug('graph pairwise coefs: \n%r', pairwise_cost)

    if np.isscalar(gc_regul) and gc_regul <= 0:
        logging.debug('gc_regul=%f so we use just argmax()', gc_regul)
        graph_labels = np.argmin(unary_cost, axis=-1).astype(np.int32)
    else:
        logging.debug('perform GraphCut')
        graph_labels = cut_general_graph(edges, edge_weights, unary_cost, pairwise_cost, algorithm='expansion',
                                         n_iter=-1)
    insert_gc_debug_images(debug_visual, segments, graph_labels, unary_cost, edges, edge_weights)
    return graph_labels


def count_label_transitions_connected_segments(dict_slics, dict_labels, nb_labels=None):
    """ count label transitions among connected superpixels over a set of images

    >>> dict_slics = {'a': np.array([[0] * 3 + [1] * 3 + [2] * 3 + [3] * 3 + [4] * 3,
    ...                              [5] * 3 + [6] * 3 + [7] * 3 + [8] * 3 + [9] * 3])}
    >>> dict_labels = {'a': np.array([0, 0, 1, 1, 2, 0, 1, 1, 0, 2])}
    >>> count_label_transitions_connected_segments(dict_slics, dict_labels).tolist()  # doctest: +SKIP
    [[2.0, 5.0, 1.0], [5.0, 3.0, 1.0], [1.0, 1.0, 1.0]]
    """
    if not nb_labels:
        uq = np.unique(np.hstack([np.unique(lbs) for lbs in dict_labels.values()]))
        nb_labels = int(np.max(uq)) + 1
    transitions = np.zeros((nb_labels, nb_labels))
    for name in dict_slics:
        if (np.max(dict_slics[name]) + 1) != len(dict_labels[name]):
            raise ValueError('dims are not matching - max slic (%i) and label (%i)' %
                             (np.max(dict_slics[name]), len(dict_labels[name])))
        _, edges = get_vertexes_edges(dict_slics[name])
        label_edges = np.asarray(dict_labels[name])[np.asarray(edges)]
        np.add.at(transitions, (label_edges[:, 0], label_edges[:, 1]), 1)
        np.add.at(transitions, (label_edges[:, 1], label_edges[:, 0]), 1)
    transitions[np.diag_indices(nb_labels)] /= 2
    return transitions


def compute_pairwise_cost_from_transitions(trans, min_prob=1e-9):
    """ pairwise cost ``log(1 / ratio)`` from label-transition counts

    >>> trans = np.array([[25., 5., 0.], [5., 10., 8.], [0., 8., 30.]])
    >>> np.round(compute_pairwise_cost_from_transitions(trans), 3).tolist()
    [[0.182, 1.526, 20.723], [1.526, 0.833, 1.056], [20.723, 1.056, 0.236]]
    """
    trans = np.asarray(trans, dtype=np.float64)
    ratio = trans / np.tile(np.sum(trans, axis=0), (len(trans), 1))
    ratio = np.maximum(ratio, ratio.T) * (1 - np.eye(len(ratio))) + np.diag(np.diag(ratio)) if ratio.ndim == 2 else ratio
    ratio[ratio < min_prob] = min_prob
    return np.log(1. / ratio)
