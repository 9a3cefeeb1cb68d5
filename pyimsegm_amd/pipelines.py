"""Pipelines: SLIC superpixels -> descriptors -> class model -> GraphCut.

Host-side mirror of the unsupervised part of the reference module ``imsegm/pipelines.py``
(``pipe_color2d_slic_features_model_graphcut`` :46, ``estim_model_classes_group`` :113,
``segment_color2d_slic_features_model_graphcut`` :160, ``compute_color2d_superpixels_features``
:244) and of its supervised part (``wrapper_compute_color2d_slic_features_labels`` :272,
``train_classif_color2d_slic_features`` :292).  One image is uploaded once; superpixels, descriptors, adjacency graph, graph cut and the
final ``proba[slic]`` / ``labels[slic]`` gathers all work on the device-resident session, only
K x F features, E edges and K x C probabilities cross the PCIe bus in between (the class model is
scikit-learn on the host, as in the reference).
"""
import logging

import numpy as np

from pyimsegm_amd import _hip
from pyimsegm_amd.descriptors import (FEATURES_SET_COLOR, _selected_features_color2d, compute_selected_features_gray3d,
                                      compute_selected_features_img2d, norm_features)
from pyimsegm_amd.graph_cuts import estim_class_model, predict_proba, segment_graph_cut_general
from pyimsegm_amd.labeling import histogram_regions_labels_norm
from pyimsegm_amd.superpixels import _open_session, _open_volume, _release_session, _run_slic, _run_slic3d

#: select basic features extracted from superpixels
FTS_SET_SIMPLE = FEATURES_SET_COLOR
#: default modeling / clustering for unsupervised segmentation
CLUSTER_METHOD = 'GMM'
#: default classifier for supervised segmentation
CLASSIF_NAME = 'RandForest'
#: number of images held out per fold of the group cross-validation (reference ``pipelines.py:38``)
CROSS_VAL_LEAVE_OUT = 2
#: default number of images a process keeps in flight on its GPU (worker threads with one HIP stream
#: each; the reference's ``NB_WORKERS`` counts pool processes, ``pipelines.py:32``)
NB_WORKERS = 2


class _ResidentImage(object):
    """one image on the device: superpixels + features, graph cut, gathers"""

    def __init__(self, image, dict_features, sp_size, sp_regul, session=None, reuse=False):
        """``session``: (Image2D, normalize_mode) of an image that is already uploaded (bench loop);
        ``reuse``: recycle the device buffers of the previous image of the same size on this thread (batches)"""
        if sp_regul <= 0.:
            raise ValueError('slic. regularisation must be positive')
        image = np.asarray(image)
        self.image = image
        logging.debug('run Superpixel clustering.')
        self.own_session = session is None
        self.reuse = reuse
        self.sess, mode = _open_session(image, reuse=reuse) if session is None else session
        self.nb_labels = _run_slic(self.sess, mode, sp_size, sp_regul)
        logging.debug('extract slic/superpixels features.')
        self._slic = None
        if image.ndim == 3 and set(dict_features) == {'color'} and image.dtype in (np.uint8, np.float64) \
                and set(dict_features['color']) <= {'mean', 'std', 'energy'}:
            # everything stays on the device: statistics of the uploaded image on the resident labels
            features, _ = _selected_features_color2d(image, None, dict_features, sess=self.sess)
        else:
            features, _ = compute_selected_features_img2d(image, self.slic, dict_features)
        features[np.isnan(features)] = 0
        self.features = features

    @property
    def slic(self):
        if self._slic is None:
            self._slic = self.sess.get_labels()
        return self._slic

    def segment(self, proba, gc_regul, gc_edge_type, debug_visual=None, classes=None, to_host=True, want_soft=True):
        image = self.image
        # the graph stage only reads the label map of the session; `segments` is passed for its
        # ndim / debug output and is only materialised on the host when somebody needs it
        segments = self.slic if (debug_visual is not None or gc_edge_type == 'color') else _ShapeOnly(self.sess.shape)
        graph_labels = segment_graph_cut_general(segments, proba, image, self.features, gc_regul, gc_edge_type,
                                                 debug_visual=debug_visual, _session=self.sess)
        if classes is not None:
            graph_labels = np.asarray(classes)[graph_labels]
        segm, segm_soft = self.sess.gather(graph_labels, proba if want_soft else None, to_host=to_host)
        if to_host and classes is not None and np.asarray(classes).dtype != np.int32:
            segm = segm.astype(np.asarray(classes).dtype)
        return segm, segm_soft

    def segment_with_model(self, model, gc_regul, gc_edge_type, host_pool, classes=None, to_host=True, want_soft=True):
        """ as ``segment(predict_proba(model, features), ...)`` with the numpy stages (class probabilities, unary /
        pairwise costs, edge weights) evaluated by a helper process of ``host_pool`` (:mod:`pyimsegm_amd.hostpool`, the
        model was handed over by ``host_pool.set_model``): identical numbers, but this thread holds the interpreter
        lock only for the glue, and the graph kernels follow the descriptor kernels without a host stage in between.
        Edge types that need the image (``'color'``) take the in-process route. """
        if gc_edge_type == 'color' or not (np.isscalar(gc_regul) and gc_regul > 0):
            return self.segment(predict_proba(model, self.features), gc_regul, gc_edge_type, classes=classes,
                                to_host=to_host, want_soft=want_soft)
        from pyimsegm_amd.graph_cuts import _edges_centres, cut_general_graph
        edges, centres, _ = _edges_centres(_ShapeOnly(self.sess.shape), self.sess)
        edges = np.array(edges, dtype=np.int32).reshape(-1, 2)
        proba, unary_cost, pairwise_cost, edge_weights = host_pool.terms(self.features, edges, centres, gc_regul,
                                                                         gc_edge_type)
        graph_labels = cut_general_graph(edges, edge_weights, unary_cost, pairwise_cost, algorithm='expansion', n_iter=-1)
        if classes is not None:
            graph_labels = np.asarray(classes)[graph_labels]
        segm, segm_soft = self.sess.gather(graph_labels, proba if want_soft else None, to_host=to_host)
        if to_host and classes is not None and np.asarray(classes).dtype != np.int32:
            segm = segm.astype(np.asarray(classes).dtype)
        return segm, segm_soft

    def mean_colour_image(self):
        """ ``skimage.color.label2rgb(slic, image, kind='avg')`` (reference ``pipelines.py:93``, the ``slic_mean`` debug
        image): per-superpixel mean colour gathered back to the pixels, on the device; as in scikit-image the
        label 0 counts as background and stays black """
        means, _, _ = self.sess.color_stats(mean=True, energy=False, var=False)
        means = np.array(means, dtype=np.float64)
        means[0] = 0.
        _, out = self.sess.gather(None, means)
        return out

    def fill_debug(self, debug_visual):
        if debug_visual is None:
            return
        image = self.image
        debug_visual['image'] = image if image.ndim == 3 else np.repeat(image[:, :, None], 3, axis=2)
        debug_visual['slic'] = self.slic
        debug_visual['slic_mean'] = self.mean_colour_image()

    def close(self):
        if self.own_session:
            if self.reuse:
                _release_session(self.sess)
            else:
                self.sess.close()


class _ShapeOnly(object):
    """stand-in for a label map whose pixels live on the device"""

    def __init__(self, shape):
        self.shape = shape
        self.ndim = len(shape)

    def __array__(self, *args, **kwargs):
        raise RuntimeError('the label map is device resident')


def compute_color2d_superpixels_features(image, dict_features, sp_size=30, sp_regul=0.2):
    """ SLIC superpixels of an image and the selected features per superpixel

    :param ndarray image: input RGB image
    :param dict(list(str)) dict_features: features to be extracted, e.g. ``{'color': ['mean']}``
    :param int sp_size: initial size of a superpixel (edge length)
    :param float sp_regul: regularisation in (0, 1): 0 elastic, 1 nearly square segments
    :return tuple(ndarray,ndarray): superpixel label map, features K x F
    """
    res = _ResidentImage(image, dict_features, sp_size, sp_regul)
    slic, features = res.slic, res.features
    res.close()
    logging.debug('list of features RAW: %r', features.shape)
    return slic, features


def wrapper_compute_color2d_slic_features_labels(img_annot, sp_size, sp_regul, dict_features, label_purity):
    """ superpixels, their features and their training labels from an annotation (reference ``pipelines.py:272-289``)

    The label of a superpixel is the annotation label that covers most of it; superpixels whose best label covers
    less than ``label_purity`` of them, or whose best label is a negative ("do not care") annotation, get -1.
    SLIC, the descriptors and the superpixel x annotation histogram all run on one device-resident session.

    :param tuple(ndarray,ndarray) img_annot: image and its annotation (integer labels, negative = ignore)
    :return tuple(ndarray,ndarray,ndarray): superpixel map, features K x F, labels K
    """
    from pyimsegm_amd.utilities import ImageDimensionError
    img, annot = img_annot
    annot = np.asarray(annot).astype(int)            # binary annotations become integer labels
    if np.shape(img)[:2] != annot.shape[:2]:
        raise ImageDimensionError('image %r and annot %r should match' % (np.shape(img), annot.shape))
    res = _ResidentImage(img, dict_features, sp_size, sp_regul)
    neg_label = np.max(annot) + 1 if np.sum(annot < 0) > 0 else None
    if neg_label is not None:
        annot[annot < 0] = neg_label
    label_hist = histogram_regions_labels_norm(None, annot, _session=res.sess)
    slic, features = res.slic, res.features
    res.close()
    labels = np.argmax(label_hist, axis=1)
    purity = np.max(label_hist, axis=1)
    if neg_label is not None:
        labels[labels == neg_label] = -1
    labels[purity < label_purity] = -1
    return slic, features, labels


def train_classif_color2d_slic_features(
    list_images,
    list_annots,
    dict_features,
    sp_size=30,
    sp_regul=0.2,
    clf_name=CLASSIF_NAME,
    label_purity=0.9,
    feature_balance='unique',
    pca_coef=None,
    nb_classif_search=1,
    nb_hold_out=CROSS_VAL_LEAVE_OUT,
    nb_workers=1,
):
    """ train a classifier on a list of annotated images (reference ``pipelines.py:292-379``)

    :param list(ndarray) list_images: RGB images
    :param list(ndarray) list_annots: annotations, integer labels (negative = do not care)
    :param dict(list(str)) dict_features: features to be extracted
    :param int sp_size: initial size of a superpixel (edge length)
    :param float sp_regul: regularisation in (0, 1)
    :param str clf_name: classifier, see :func:`classification.create_classifiers`
    :param float label_purity: minimal share of a superpixel its label has to cover
    :param str feature_balance: how to balance the training set (per image): 'unique', 'random', 'kmeans' or None
    :param float pca_coef: PCA coefficient or None
    :param int nb_classif_search: number of tries of the hyper-parameter search (<= 1: train once)
    :param int nb_hold_out: images held out per cross-validation fold of that search
    :param int nb_workers: images in flight on this GPU (worker threads) and jobs of the search
    :return tuple(obj,list(ndarray),list(ndarray),list(ndarray)): classifier, superpixel maps, features, labels
    """
    from pyimsegm_amd.classification import (CrossValidateGroups, convert_set_features_labels_2_dataset,
                                             create_classif_search_train_export)
    logging.info('TRAIN Superpixels-Features-Classifier')
    if len(list_images) != len(list_annots):
        raise ValueError('size of images (%i) and annotations (%i) should match' % (len(list_images), len(list_annots)))

    def _compute(img_annot):
        return wrapper_compute_color2d_slic_features_labels(img_annot, sp_size, sp_regul, dict_features, label_purity)

    pairs = list(zip(list_images, list_annots))
    if nb_workers and nb_workers > 1 and len(pairs) > 1:
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=int(nb_workers)) as pool:
            results = list(pool.map(_compute, pairs))
    else:
        results = [_compute(pair) for pair in pairs]
    list_slic = [r[0] for r in results]
    list_features = [r[1] for r in results]
    list_labels = [r[2] for r in results]

    # one training set over all images; the "do not care" superpixels (-1) are dropped
    features, labels, sizes = convert_set_features_labels_2_dataset(
        dict(zip(range(len(list_features)), list_features)), dict(zip(range(len(list_labels)), list_labels)),
        balance_type=feature_balance, drop_labels=[-1])
    features = np.nan_to_num(features)
    # hold out whole images when there are enough of them, else plain 10-fold
    cv = CrossValidateGroups(sizes, nb_hold_out=nb_hold_out) if len(sizes) > (nb_hold_out * 5) else 10
    classif, _ = create_classif_search_train_export(clf_name, features, labels, pca_coef=pca_coef, cross_val=cv,
                                                    nb_search_iter=nb_classif_search, nb_workers=nb_workers)
    return classif, list_slic, list_features, list_labels


def pipe_color2d_slic_features_model_graphcut(
    image,
    nb_classes,
    dict_features,
    sp_size=30,
    sp_regul=0.2,
    pca_coef=None,
    use_scaler=True,
    estim_model='GMM',
    gc_regul=1.,
    gc_edge_type='model',
    debug_visual=None,
):
    """ complete unsupervised pipeline: superpixels, features, mixture model, GraphCut

    :param ndarray image: input RGB image
    :param int nb_classes: number of classes to be segmented (indexing from 0)
    :param dict dict_features: {clr: list(str)}
    :param int sp_size: initial size of a superpixel (edge length)
    :param float sp_regul: regularisation in (0, 1)
    :param float pca_coef: range (0, 1) or None
    :param bool use_scaler: use a standard scaler in front of the model
    :param str estim_model: estimating model, see :func:`graph_cuts.estim_class_model`
    :param float gc_regul: GraphCut regularisation
    :param str gc_edge_type: GraphCut edge type
    :param dict debug_visual: filled with intermediate results if given
    :return tuple(ndarray,ndarray): segmentation H x W, soft segmentation H x W x nb_classes

    >>> np.random.seed(0)
    >>> image = np.random.random((125, 150, 3)) / 2.
    >>> image[:, :75] += 0.5
    >>> segm, seg_soft = pipe_color2d_slic_features_model_graphcut(image, 2, {'color': ['mean']})  # doctest: +SKIP
    >>> segm.shape  # doctest: +SKIP
    (125, 150)
    """
    logging.info('PIPELINE Superpixels-Features-GMM-GraphCut')
    res = _ResidentImage(image, dict_features, sp_size, sp_regul)
    res.fill_debug(debug_visual)
    model = estim_class_model(res.features, nb_classes, estim_model, pca_coef, use_scaler)
    proba = model.predict_proba(res.features)
    logging.debug('list of probabilities: %r', proba.shape)
    segm, segm_soft = res.segment(proba, gc_regul, gc_edge_type, debug_visual)
    res.close()
    return segm, segm_soft


def estim_model_classes_group(
    list_images,
    nb_classes,
    dict_features,
    sp_size=30,
    sp_regul=0.2,
    use_scaler=True,
    pca_coef=None,
    model_type='GMM',
    nb_workers=NB_WORKERS,
):
    """ estimate one class model from the superpixel features of a sequence of images

    :return tuple(model, list(ndarray)): fitted scikit-learn pipeline, features per image
    """
    def _features(image):
        return compute_color2d_superpixels_features(image, dict_features, sp_size=sp_size, sp_regul=sp_regul)[1]

    if nb_workers and nb_workers > 1 and len(list_images) > 1:
        # several images in flight on this GPU: worker threads, one HIP stream each
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=int(nb_workers)) as pool:
            list_features = list(pool.map(_features, list_images))
    else:
        list_features = [_features(image) for image in list_images]
    features = np.nan_to_num(np.concatenate(tuple(list_features), axis=0))
    model = estim_class_model(features, nb_classes, model_type, pca_coef, use_scaler)
    return model, list_features


def segment_color2d_slic_features_model_graphcut(
    image,
    model_pipeline,
    dict_features,
    sp_size=30,
    sp_regul=0.2,
    gc_regul=1.,
    gc_edge_type='model',
    debug_visual=None,
):
    """ segmentation with a given (pre-trained) model: superpixels, features, predict, GraphCut

    :param ndarray image: input RGB image
    :param obj model_pipeline: fitted model with ``predict_proba``
    :return tuple(ndarray,ndarray): segmentation H x W, soft segmentation H x W x nb_classes
    """
    logging.info('PIPELINE Superpixels-Features-Model-GraphCut')
    res = _ResidentImage(image, dict_features, sp_size, sp_regul)
    res.fill_debug(debug_visual)
    proba = predict_proba(model_pipeline, res.features)
    logging.debug('list of probabilities: %r', proba.shape)
    classes = getattr(model_pipeline, 'classes_', None)
    segm, segm_soft = res.segment(proba, gc_regul, gc_edge_type, debug_visual, classes=classes)
    res.close()
    return segm, segm_soft


def segment_batch_color2d_slic_features_model_graphcut(list_images, model_pipeline, dict_features, sp_size=30,
                                                       sp_regul=0.2, gc_regul=1., gc_edge_type='model', group=None,
                                                       nb_workers=NB_WORKERS, host_procs=True):
    """ segment a batch of equally-sized images with a given model, sharded over the GPUs of one node

    Multi-GPU counterpart of mapping :func:`segment_color2d_slic_features_model_graphcut` over a
    process pool (reference ``run_segm_slic_model_graphcut.py:505-514``): one process per GPU
    (``torchrun``), image *i* goes to rank ``i mod world``, the label maps are gathered on rank 0
    over RCCL.  Without a process group it processes the images on this GPU, ``nb_workers`` of them
    in flight at a time.

    :return list(ndarray): label maps on rank 0 (``None`` on the other ranks)
    """
    from pyimsegm_amd.distributed import Group, segment_batch_sharded
    own = group is None
    if own:
        group = Group()

    classes = getattr(model_pipeline, 'classes_', None)
    # several images in flight: their numpy stages (class model, graph-cut terms) run in helper processes, so the
    # worker threads do not queue for the interpreter lock (pyimsegm_amd/hostpool.py; same functions, same numbers)
    host_pool = None
    if host_procs and nb_workers and nb_workers > 1 and len(list_images) > 1:
        from pyimsegm_amd.hostpool import shared_pool
        try:
            host_pool = shared_pool(nb_workers)      # started once per process, reused by later batches
            host_pool.set_model(model_pipeline)
        except Exception as ex:                      # e.g. a model that cannot be pickled: stay in process
            logging.warning('host helper processes not available (%s): class model evaluated in the worker threads', ex)
            host_pool = None

    def _segment(image):
        # as segment_color2d_slic_features_model_graphcut, minus what a batch does not need: the soft
        # segmentation stays on the device and the session buffers are recycled from image to image
        res = _ResidentImage(image, dict_features, sp_size, sp_regul, reuse=True)
        if host_pool is not None:
            segm, _ = res.segment_with_model(model_pipeline, gc_regul, gc_edge_type, host_pool, classes=classes,
                                             want_soft=False)
        else:
            proba = predict_proba(model_pipeline, res.features)
            segm, _ = res.segment(proba, gc_regul, gc_edge_type, classes=classes, want_soft=False)
        res.close()
        return segm

    out = segment_batch_sharded(list_images, _segment, group, nb_workers=nb_workers)
    if own:
        group.close()
    return out


def pipe_gray3d_slic_features_model_graphcut(
    image,
    nb_classes,
    dict_features,
    spacing=(12, 1, 1),
    sp_size=15,
    sp_regul=0.2,
    gc_regul=0.1,
):
    """ complete pipeline on a gray volume: supervoxels, features, mixture model, GraphCut
    (reference ``pipelines.py:382-431``)

    The volume is uploaded once; supervoxels, connected-component relabelling, gray statistics, the
    6-connected adjacency graph, the graph cut and the final ``graph_labels[slic]`` gather run on the
    device-resident session.

    :param ndarray image: input gray volume D x H x W
    :param int nb_classes: number of classes to be segmented (indexing from 0)
    :param dict(list(str)) dict_features: features to be extracted, e.g. ``{'color': ['mean']}``
    :param tuple(int,int,int) spacing: voxel spacing
    :param int sp_size: initial size of a supervoxel (edge length)
    :param float sp_regul: regularisation in (0, 1): 0 elastic, 1 nearly cubic segments
    :param float gc_regul: GraphCut regularisation
    :return ndarray: int32 class per voxel, D x H x W

    >>> np.random.seed(0)
    >>> image = np.random.random((5, 125, 150)) / 2.
    >>> image[:, :, :75] += 0.5
    >>> segm = pipe_gray3d_slic_features_model_graphcut(image, 2, {'color': ['mean']})  # doctest: +SKIP
    >>> segm.shape  # doctest: +SKIP
    (5, 125, 150)
    """
    logging.info('PIPELINE Superpixels-Features-GraphCut')
    image = np.asarray(image)
    sess = _open_volume(image)
    _run_slic3d(sess, sp_size, sp_regul, spacing)
    logging.info('extract segments/superpixels features.')
    slic = None
    resident = set(dict_features) == {'color'} and set(dict_features['color']) <= {'mean', 'std', 'energy'} \
        and (image.dtype.kind != 'f' or bool(np.isfinite(image.sum(dtype=np.float64))))    # one pass, no temporaries
    if resident:
        features, _ = compute_selected_features_gray3d(image, _ShapeOnly(sess.shape), dict_features, sess=sess)
    else:
        slic = sess.get_labels()
        features, _ = compute_selected_features_gray3d(image, slic, dict_features)
    logging.debug('list of features RAW: %r', features.shape)
    features[np.isnan(features)] = 0

    logging.info('norm all features.')
    features, _ = norm_features(features)
    logging.debug('list of features NORM: %r', features.shape)

    model = estim_class_model(features, nb_classes)
    proba = model.predict_proba(features)
    logging.debug('list of probabilities: %r', proba.shape)

    graph_labels = segment_graph_cut_general(_ShapeOnly(sess.shape), proba, image, features, gc_regul, _session=sess)
    segm, _ = sess.gather(graph_labels)
    sess.close()
    return segm
