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
                                      compute_selected_features_img2d, norm_features, resident_feature_groups,
                                      resident_feature_table)
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


_device_models = None


def _device_gmm(model):
    """:class:`_hip.DeviceGmm` of a fitted ``Pipeline([StandardScaler,] GaussianMixture('full'))`` (cached per model
    object as long as its parameters are the same arrays), or None when the model has to be evaluated by scikit-learn"""
    global _device_models
    import weakref
    if _device_models is None:
        _device_models = weakref.WeakKeyDictionary()
    try:
        last = model.steps[-1][1] if hasattr(model, 'steps') else model
        chol = getattr(last, 'precisions_cholesky_', None)
        hit = _device_models.get(model)
        if hit is not None and hit[0] is chol:
            return hit[1]
        dev = _hip.DeviceGmm(model)
        _device_models[model] = (chol, dev)
        return dev
    except TypeError:
        return None
    except Exception as ex:       # private scikit-learn layout changed ...: the host evaluates the model
        logging.debug('class model stays on the host: %s', ex)
        return None


def _segment_color2d_one_call(image, model, dict_features, sp_size, sp_regul, gc_regul, gc_edge_type, want_soft=True, reuse=False,
                              with_session=None):
    """``segment_color2d_slic_features_model_graphcut`` as ONE call into the library (:meth:`_hip.Image2D.run_color`) when
    everything it needs lives on the device: uint8 / float64 colour image, colour mean / std / energy features, a
    scaler + full-covariance mixture, an edge type the device evaluates.  Returns None when the general path is needed.
    ``with_session(sess)`` is called before the session is recycled (the label map is still in its HBM buffer then)."""
    from pyimsegm_amd.graph_cuts import compute_pairwise_cost
    from pyimsegm_amd.superpixels import SLIC_MAX_ITER, SLIC_START_LABEL, _slic_params
    image = np.asarray(image)
    flags = dict_features.get('color', ()) if set(dict_features) == {'color'} else None
    if image.ndim != 3 or image.shape[2] != 3 or image.dtype not in (np.uint8, np.float64) or not flags \
            or not set(flags) <= {'mean', 'std', 'energy'} or gc_edge_type not in _hip.EDGE_TYPES:
        return None
    if sp_regul <= 0.:
        raise ValueError('slic. regularisation must be positive')
    gmm = _device_gmm(model)
    if gmm is None or gmm.n_features != 3 * len(set(flags)):
        return None
    if image.dtype != np.uint8 and not bool(np.isfinite(image.sum(dtype=np.float64))):
        return None
    n_seg, compact = _slic_params(image.shape[:2], sp_size, sp_regul)
    if n_seg < 1:
        raise ValueError('superpixel size %r is larger than the image %r' % (sp_size, image.shape[:2]))
    pairwise = compute_pairwise_cost(gc_regul, (0, gmm.n_classes))
    classes = getattr(model, 'classes_', None)
    sess = None
    if reuse:
        sess = _hip.default_context().idle_sessions.pop(image.shape[:2], None)
    if sess is None:
        sess = _hip.Image2D(image.shape[0], image.shape[1])
    try:
        segm, soft = sess.run_color(image, n_seg, compact, gmm, pairwise, gc_edge_type,
                                    ('mean' in flags, 'std' in flags, 'energy' in flags), max_iter=SLIC_MAX_ITER,
                                    start_label=SLIC_START_LABEL, use_graphcut=not (np.isscalar(gc_regul) and gc_regul <= 0),
                                    classes=None if classes is None else np.asarray(classes).astype(np.int32), want_soft=want_soft)
        if with_session is not None:
            with_session(sess)
    finally:
        if reuse:
            _release_session(sess)
        else:
            sess.close()
    if classes is not None and np.asarray(classes).dtype != np.int32:
        segm = segm.astype(np.asarray(classes).dtype)
    return segm, soft


#: images per launch chain of the batch path (csrc/batch.hip); 0 switches it off (one image per call, worker threads)
BATCH_IMAGES = 8


def _segment_color2d_batch_call(images, model, dict_features, sp_size, sp_regul, gc_regul, gc_edge_type, to_host=True, out=None,
                                with_batch=None):
    """``segment_color2d_slic_features_model_graphcut`` of SEVERAL images of one size and dtype as ONE call into the library
    (:meth:`_hip.Batch2D.run_color`: every kernel launched once for the batch, image = blockIdx.z) -- under the conditions
    of :func:`_segment_color2d_one_call`.  Returns the list of segmentations, or None when the images have to go one by one.
    The batch object is kept by the thread's context for the next batch of the same size; ``with_batch(batch)`` is called
    before it is given back (the maps are still in its HBM buffers then)."""
    from pyimsegm_amd.graph_cuts import compute_pairwise_cost
    from pyimsegm_amd.superpixels import SLIC_MAX_ITER, SLIC_START_LABEL, _slic_params
    images = [np.asarray(im) for im in images]
    flags = dict_features.get('color', ()) if set(dict_features) == {'color'} else None
    first = images[0]
    if first.ndim != 3 or first.shape[2] != 3 or first.dtype not in (np.uint8, np.float64) or not flags \
            or not set(flags) <= {'mean', 'std', 'energy'} or gc_edge_type not in _hip.EDGE_TYPES \
            or any(im.shape != first.shape or im.dtype != first.dtype for im in images):
        return None
    if sp_regul <= 0.:
        raise ValueError('slic. regularisation must be positive')
    gmm = _device_gmm(model)
    if gmm is None or gmm.n_features != 3 * len(set(flags)):
        return None
    if first.dtype != np.uint8 and not all(bool(np.isfinite(im.sum(dtype=np.float64))) for im in images):
        return None
    n_seg, compact = _slic_params(first.shape[:2], sp_size, sp_regul)
    if n_seg < 1:
        raise ValueError('superpixel size %r is larger than the image %r' % (sp_size, first.shape[:2]))
    if first.shape[0] * first.shape[1] < 4 * n_seg:          # (superpixels of a few pixels: the single-image calls take them)
        return None
    pairwise = compute_pairwise_cost(gc_regul, (0, gmm.n_classes))
    classes = getattr(model, 'classes_', None)
    ctx = _hip.default_context()
    key = ('batch', max(len(images), BATCH_IMAGES)) + first.shape[:2]
    batch = ctx.idle_sessions.pop(key, None)
    if batch is None:
        batch = _hip.Batch2D(key[1], first.shape[0], first.shape[1])
    try:
        segm = batch.run_color(images, n_seg, compact, gmm, pairwise, gc_edge_type,
                               ('mean' in flags, 'std' in flags, 'energy' in flags), max_iter=SLIC_MAX_ITER, start_label=SLIC_START_LABEL,
                               use_graphcut=not (np.isscalar(gc_regul) and gc_regul <= 0),
                               classes=None if classes is None else np.asarray(classes).astype(np.int32), to_host=to_host, out=out)
        if with_batch is not None:
            with_batch(batch)
    except BaseException:
        batch.close()
        raise
    if key in ctx.idle_sessions:
        batch.close()
    else:
        ctx.idle_sessions[key] = batch
    if segm is not None and classes is not None and np.asarray(classes).dtype != np.int32:
        segm = [a.astype(np.asarray(classes).dtype) for a in segm]
    return segm if segm is not None else []


def _segment_images_batched(images, model, dict_features, sp_size, sp_regul, gc_regul, gc_edge_type, nb_workers=1):
    """segmentations of ``images`` with BATCH_IMAGES of them per call of :func:`_segment_color2d_batch_call`, ``nb_workers``
    such calls in flight (worker threads, one HIP stream each); None when the images do not qualify (sizes, dtype, features,
    class model: see there) and have to go one by one"""
    if not images:
        return []
    chunks = [list(range(lo, min(lo + BATCH_IMAGES, len(images)))) for lo in range(0, len(images), BATCH_IMAGES)]

    def run(chunk):
        return _segment_color2d_batch_call([images[i] for i in chunk], model, dict_features, sp_size, sp_regul, gc_regul, gc_edge_type)

    first = run(chunks[0])
    if first is None:
        return None
    parts = [first]
    if len(chunks) > 1:
        if nb_workers and nb_workers > 1 and len(chunks) > 2:
            from concurrent.futures import ThreadPoolExecutor
            with ThreadPoolExecutor(max_workers=int(nb_workers)) as pool:
                parts += list(pool.map(run, chunks[1:]))
        else:
            parts += [run(chunk) for chunk in chunks[1:]]
    if any(part is None for part in parts):
        return None
    return [np.array(segm) for part in parts for segm in part]      # (copies: the page-locked result arrays go back to their pool)


class _ResidentImage(object):
    """one image on the device: superpixels + features, then the fused class model / graph cut / gathers"""

    def __init__(self, image, dict_features, sp_size, sp_regul, session=None, reuse=False, features_to_host=True):
        """``session``: (Image2D, normalize_mode) of an image that is already uploaded (bench loop);
        ``reuse``: recycle the device buffers of the previous image of the same size on this thread (batches);
        ``features_to_host``: False when nobody on the host needs the K x F table (pre-fitted mixture on the device)"""
        if sp_regul <= 0.:
            raise ValueError('slic. regularisation must be positive')
        image = np.asarray(image)
        self.image = image
        logging.debug('run Superpixel clustering.')
        self.own_session = session is None
        self.reuse = reuse
        self.sess, mode = _open_session(image, reuse=reuse) if session is None else session
        try:
            self.nb_labels = _run_slic(self.sess, mode, sp_size, sp_regul)
            logging.debug('extract slic/superpixels features.')
            self._slic = None
            self._features = None
            self.resident_features = False
            self._table_columns = 0
            groups = resident_feature_groups(dict_features) if image.ndim == 3 and image.dtype in (np.uint8, np.float64) else None
            # everything stays on the device: colour statistics of the uploaded image / Leung-Malik statistics of its filter
            # responses on the resident labels, side by side in the feature table the class model and the 'features' edge type read
            # (float images: NaN / inf would have to be replaced first, descriptors.py:818 -- checked on the host)
            if groups is not None and (image.dtype == np.uint8 or bool(np.isfinite(image.sum(dtype=np.float64)))):
                self._table_columns = resident_feature_table(self.sess, groups)
                self.resident_features = True
                if features_to_host:
                    self._features = self.sess.get_features(self._table_columns)
            if not self.resident_features:
                features, _ = compute_selected_features_img2d(image, self.slic, dict_features)
                features[np.isnan(features)] = 0
                self._features = features
        except BaseException:
            self.close()
            raise

    @property
    def features(self):
        if self._features is None:        # resident table that was not needed on the host so far
            self._features = self.sess.get_features(self._table_columns)
        return self._features

    @property
    def slic(self):
        if self._slic is None:
            self._slic = self.sess.get_labels()
        return self._slic

    def segment(self, proba, gc_regul, gc_edge_type, debug_visual=None, classes=None, to_host=True, want_soft=True,
                model=None, segm_dtype=None, soft_dtype=None):
        """ graph cut + gathers.  ``proba``: K x C from the host, or None with ``model`` = a mixture the device evaluates
        on the resident features.  Everything runs in one fused call (:meth:`_hip.Image2D.segment`) unless the edge
        type needs the image on the host (``'color'``). """
        image = self.image
        gmm = None
        if proba is None:
            gmm = _device_gmm(model) if self.resident_features else None
            if gmm is None:
                proba = predict_proba(model, self.features)
        nb_classes = gmm.n_classes if gmm is not None else np.shape(proba)[1]
        if gc_edge_type in _hip.EDGE_TYPES and (gc_edge_type != 'features' or self.resident_features):
            from pyimsegm_amd.graph_cuts import compute_pairwise_cost, insert_gc_debug_images
            pairwise = compute_pairwise_cost(gc_regul, (self.nb_labels, nb_classes))
            use_gc = not (np.isscalar(gc_regul) and gc_regul <= 0)
            cls = None if classes is None else np.asarray(classes)
            res = self.sess.segment(pairwise, gc_edge_type, gmm=gmm, proba=proba, use_graphcut=use_gc,
                                    classes=None if cls is None else cls.astype(np.int32), want_segm=to_host,
                                    want_soft=to_host and want_soft, debug=debug_visual is not None,
                                    keep_soft_on_device=want_soft and not to_host, segm_dtype=segm_dtype, soft_dtype=soft_dtype)
            if debug_visual is not None:
                insert_gc_debug_images(debug_visual, self.slic, res['graph_labels'], res['unary'], res['edges'],
                                       res['edge_weights'])
            segm, segm_soft = res.get('segm'), res.get('soft')
        else:
            # the graph stage only reads the label map of the session; `segments` is passed for its
            # ndim / debug output and is only materialised on the host when somebody needs it
            segments = self.slic if (debug_visual is not None or gc_edge_type == 'color') else _ShapeOnly(self.sess.shape)
            graph_labels = segment_graph_cut_general(segments, proba, image, self.features, gc_regul, gc_edge_type,
                                                     debug_visual=debug_visual, _session=self.sess)
            if classes is not None:
                graph_labels = np.asarray(classes)[graph_labels]
            segm, segm_soft = self.sess.gather(graph_labels, proba if want_soft else None, to_host=to_host)
            if to_host and segm_dtype is not None:
                segm = segm.astype(segm_dtype)
            if to_host and soft_dtype is not None and segm_soft is not None:
                segm_soft = segm_soft.astype(soft_dtype)
        if to_host and segm_dtype is None and classes is not None and np.asarray(classes).dtype != np.int32:
            segm = segm.astype(np.asarray(classes).dtype)
        return segm, segm_soft

    def mean_colour_image(self):
        """ ``skimage.color.label2rgb(slic, image, kind='avg')`` (reference ``pipelines.py:93``, the ``slic_mean`` debug
        image): per-superpixel mean colour gathered back to the pixels, on the device;
        in scikit-image 0.18 -- the release line the reference's calls need -- ``bg_label`` is -1 for ``kind='avg'``, so label 0
        gets its mean colour like every other label (tests/golden/label2rgb.npz, from the real scikit-image) """
        means, _, _ = self.sess.color_stats(mean=True, energy=False, var=False)
        _, out = self.sess.gather(None, np.array(means, dtype=np.float64))
        return out

    def fill_debug(self, debug_visual):
        if debug_visual is None:
            return
        image = self.image
        debug_visual['image'] = image if image.ndim == 3 else np.repeat(image[:, :, None], 3, axis=2)
        debug_visual['slic'] = self.slic
        debug_visual['slic_mean'] = self.mean_colour_image()

    def close(self):
        if self.own_session and self.sess is not None:
            if self.reuse:
                _release_session(self.sess)
            else:
                self.sess.close()
            self.sess = None


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
    try:
        slic, features = res.slic, res.features
    finally:
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
    try:
        neg_label = np.max(annot) + 1 if np.sum(annot < 0) > 0 else None
        if neg_label is not None:
            annot[annot < 0] = neg_label
        label_hist = histogram_regions_labels_norm(None, annot, _session=res.sess)
        slic, features = res.slic, res.features
    finally:
        res.close()
    labels = np.argmax(label_hist, axis=1)
    purity = np.max(label_hist, axis=1)
    if neg_label is not None:
        labels[labels == neg_label] = -1
    labels[purity < label_purity] = -1
    return slic, features, labels


def train_classif_color2d_slic_features(list_images, list_annots, dict_features, sp_size=30, sp_regul=0.2,
                                        clf_name=CLASSIF_NAME, label_purity=0.9, feature_balance='unique', pca_coef=None,
                                        nb_classif_search=1, nb_hold_out=CROSS_VAL_LEAVE_OUT, nb_workers=1):
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


def pipe_color2d_slic_features_model_graphcut(image, nb_classes, dict_features, sp_size=30, sp_regul=0.2, pca_coef=None,
                                              use_scaler=True, estim_model='GMM', gc_regul=1., gc_edge_type='model',
                                              debug_visual=None):
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
    """
    logging.info('PIPELINE Superpixels-Features-GMM-GraphCut')
    res = _ResidentImage(image, dict_features, sp_size, sp_regul)
    try:
        res.fill_debug(debug_visual)
        model = estim_class_model(res.features, nb_classes, estim_model, pca_coef, use_scaler)
        # the fitted mixture is evaluated on the device (resident features) when it is the scaler + GMM pipeline,
        # by scikit-learn otherwise
        segm, segm_soft = res.segment(None, gc_regul, gc_edge_type, debug_visual, model=model)
    finally:
        res.close()
    return segm, segm_soft


def estim_model_classes_group(list_images, nb_classes, dict_features, sp_size=30, sp_regul=0.2, use_scaler=True, pca_coef=None,
                              model_type='GMM', nb_workers=NB_WORKERS, group=None):
    """ estimate one class model from the superpixel features of a sequence of images (reference ``pipelines.py:113-157``)

    With a multi-rank ``group`` (:class:`pyimsegm_amd.distributed.Group`) every rank extracts the features of its images
    ``i = rank, rank + world, ...``, the K_i x F blocks are gathered on rank 0 in image order, the model is fitted
    there once and broadcast to all ranks.

    :return tuple(model, list(ndarray)): fitted scikit-learn pipeline, features per image (all of them on rank 0)
    """
    def _features(image):
        return compute_color2d_superpixels_features(image, dict_features, sp_size=sp_size, sp_regul=sp_regul)[1]

    def _fit(features):
        return estim_class_model(features, nb_classes, model_type, pca_coef, use_scaler)

    if group is not None and group.world > 1:
        from pyimsegm_amd.distributed import estim_model_classes_group_sharded
        return estim_model_classes_group_sharded(list_images, _features, _fit, group)
    if nb_workers and nb_workers > 1 and len(list_images) > 1:
        # several images in flight on this GPU: worker threads, one HIP stream each
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=int(nb_workers)) as pool:
            list_features = list(pool.map(_features, list_images))
    else:
        list_features = [_features(image) for image in list_images]
    return _fit(np.nan_to_num(np.concatenate(tuple(list_features), axis=0))), list_features


def segment_color2d_slic_features_model_graphcut(image, model_pipeline, dict_features, sp_size=30, sp_regul=0.2, gc_regul=1.,
                                                 gc_edge_type='model', debug_visual=None, segm_dtype=None, soft_dtype=None):
    """ segmentation with a given (pre-trained) model: superpixels, features, predict, GraphCut

    :param ndarray image: input RGB image
    :param obj model_pipeline: fitted model with ``predict_proba``
    :param segm_dtype: (not in the reference) ``np.uint8``: the class map leaves the device as bytes (classes < 256) -- a quarter
        of the transfer; None: int32 as the reference returns it
    :param soft_dtype: (not in the reference) ``np.float32``: the soft segmentation leaves the device as float32 (half of the
        100 MB a 2048 x 2048 x 3 float64 array takes), ``False``: it is not produced at all; None: float64 as the reference
    :return tuple(ndarray,ndarray): segmentation H x W, soft segmentation H x W x nb_classes
    """
    logging.info('PIPELINE Superpixels-Features-Model-GraphCut')
    narrow = segm_dtype is not None or soft_dtype is not None
    if debug_visual is None and not narrow:
        fast = _segment_color2d_one_call(image, model_pipeline, dict_features, sp_size, sp_regul, gc_regul, gc_edge_type)
        if fast is not None:
            return fast
    res = _ResidentImage(image, dict_features, sp_size, sp_regul, features_to_host=_device_gmm(model_pipeline) is None)
    try:
        res.fill_debug(debug_visual)
        classes = getattr(model_pipeline, 'classes_', None)
        segm, segm_soft = res.segment(None, gc_regul, gc_edge_type, debug_visual, classes=classes, model=model_pipeline,
                                      want_soft=soft_dtype is not False, segm_dtype=segm_dtype,
                                      soft_dtype=None if soft_dtype is False else soft_dtype)
    finally:
        res.close()
    return segm, segm_soft


def segment_batch_color2d_slic_features_model_graphcut(list_images, model_pipeline, dict_features, sp_size=30,
                                                       sp_regul=0.2, gc_regul=1., gc_edge_type='model', group=None,
                                                       nb_workers=NB_WORKERS):
    """ segment a batch of equally-sized images with a given model, sharded over the GPUs of one node

    Multi-GPU counterpart of mapping :func:`segment_color2d_slic_features_model_graphcut` over a
    process pool (reference ``run_segm_slic_model_graphcut.py:505-514``): one process per GPU, image *i* goes to
    rank ``i mod world``, the label maps are gathered on rank 0 over RCCL.  Without a process group it processes the
    images on this GPU, ``nb_workers`` of them in flight at a time (worker threads with one HIP stream each; a thread
    spends its time inside two C calls per image, outside the interpreter lock).

    :return list(ndarray): label maps on rank 0 (``None`` on the other ranks)
    """
    from pyimsegm_amd.distributed import Group, segment_batch_sharded
    own = group is None
    if own:
        group = Group()
    classes = getattr(model_pipeline, 'classes_', None)
    on_device = _device_gmm(model_pipeline) is not None
    if group.rccl is None and BATCH_IMAGES > 0 and on_device:
        # the images of this rank several at a time through ONE launch chain each (csrc/batch.hip); with several ranks and no
        # RCCL communicator (CPU tests, ranks sharing a GPU) the finished maps then travel over the host plane as before
        mine = group.shard(len(list_images))
        ready = _segment_images_batched([list_images[i] for i in mine], model_pipeline, dict_features, sp_size, sp_regul, gc_regul,
                                        gc_edge_type, nb_workers)
        if ready is not None:
            table = {id(list_images[i]): segm for i, segm in zip(mine, ready)}
            try:
                return segment_batch_sharded(list_images, lambda image: table[id(image)], group, nb_workers=1)
            finally:
                if own:
                    group.close()

    def _segment(image):
        # as segment_color2d_slic_features_model_graphcut, minus what a batch does not need: the soft
        # segmentation is not computed and the session buffers are recycled from image to image
        fast = _segment_color2d_one_call(image, model_pipeline, dict_features, sp_size, sp_regul, gc_regul, gc_edge_type,
                                         want_soft=False, reuse=True)
        if fast is not None:
            return fast[0]
        res = _ResidentImage(image, dict_features, sp_size, sp_regul, reuse=True, features_to_host=not on_device)
        try:
            segm, _ = res.segment(None, gc_regul, gc_edge_type, classes=classes, want_soft=False, model=model_pipeline)
        finally:
            res.close()
        return segm

    class _OnDevice(object):
        """a finished image whose label map is still in its session's HBM buffer (RCCL gather, zero copy)"""

        def __init__(self, image):
            self.res = _ResidentImage(image, dict_features, sp_size, sp_regul, reuse=True, features_to_host=not on_device)
            try:
                self.res.segment(None, gc_regul, gc_edge_type, classes=classes, to_host=False, want_soft=False,
                                 model=model_pipeline)
                self.ctx = self.res.sess.ctx
                self.device_ptr = _hip.segm_device_array(self.res.sess).__cuda_array_interface__['data'][0]
            except BaseException:
                self.res.close()
                raise

        def close(self):
            self.res.close()

    try:
        return segment_batch_sharded(list_images, _segment, group, nb_workers=nb_workers, segment_device_fn=_OnDevice)
    finally:
        if own:
            group.close()


def pipe_gray3d_slic_features_model_graphcut(image, nb_classes, dict_features, spacing=(12, 1, 1), sp_size=15, sp_regul=0.2,
                                             gc_regul=0.1):
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
    """
    logging.info('PIPELINE Superpixels-Features-GraphCut')
    image = np.asarray(image)
    sess = _open_volume(image, reuse=True)
    try:
        segm = _gray3d_on_session(sess, image, nb_classes, dict_features, spacing, sp_size, sp_regul, gc_regul)
    except Exception:
        sess.close()
        raise
    _release_session(sess)                  # (kept for the next volume of this shape: see superpixels._open_volume)
    return segm


def _touched_result(shape, dtype=np.int32, threads=4):
    """(array, join): a fresh pageable result array whose pages worker threads touch while the caller goes on -- a download into
    untouched pages runs at 25 GB/s, into touched ones at the 55 GB/s of the link (tools/micro/pageable_copy.py): for the 4.3 GB
    label map of a 64 x 4096 x 4096 volume 170 against 80 ms, hidden behind the SLIC sweeps and the model fit.  Small arrays are
    returned as they are."""
    import threading
    out = np.empty(shape, dtype=dtype)
    if out.nbytes < (64 << 20):
        return out, (lambda: None)
    flat = out.reshape(-1).view(np.uint8)
    edges = np.linspace(0, flat.size, threads + 1).astype(np.int64) // 4096 * 4096
    edges[-1] = flat.size

    def touch(a, b):
        flat[a:b:4096] = 0                 # one byte per page (the array's content is overwritten by the download)

    workers = [threading.Thread(target=touch, args=(int(edges[i]), int(edges[i + 1])), daemon=True) for i in range(threads)]
    for w in workers:
        w.start()
    return out, (lambda: [w.join() for w in workers])


def _result_array(shape, dtype=np.int32):
    """(array, wait): where the class map of a volume goes.  64 MB and more: a PAGE-LOCKED array out of the recycling pool of
    ``_hip.pinned_empty`` (round 6) -- a fresh pageable array of 4.3 GB costs its page faults on the way in (hidden behind the
    sweeps by :func:`_touched_result`) and 0.25 s of ``munmap`` when the caller lets go of it, every call; the page-locked block
    returns to the pool instead and serves the next call, and the download runs as plain DMA.  When the host refuses that much
    page-locked memory: the touched pageable array."""
    nbytes = int(np.prod(shape, dtype=np.int64)) * np.dtype(dtype).itemsize
    if nbytes >= (64 << 20):
        try:
            return _hip.pinned_empty(shape, dtype), (lambda: None)
        except (_hip.HipError, _hip.HipUnavailableError):
            pass
    return _touched_result(shape, dtype)


def _gray3d_on_session(sess, image, nb_classes, dict_features, spacing, sp_size, sp_regul, gc_regul):
    segm_buf, segm_ready = _result_array(sess.shape)
    _run_slic3d(sess, sp_size, sp_regul, spacing)
    logging.info('extract segments/superpixels features.')
    slic = None
    resident = set(dict_features) == {'color'} and set(dict_features['color']) <= {'mean', 'std', 'energy'} \
        and (image.dtype.kind != 'f' or (sess.all_finite() if image.dtype == sess.dtype
                                          else bool(np.isfinite(image.sum(dtype=np.float64)))))    # (asked on the device)
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

    # the graph of the supervoxels (neighbour pairs, centres, edges, arcs) depends on the label map alone: it is enqueued HERE and
    # built by the device while the host fits the mixture -- the fused call below finds it ready
    fused = True
    try:
        sess.graph_prepare()
    except _hip.HipFusedPathError as ex:
        fused = False
        logging.info('volume graph by neighbour tables (%s)', ex)

    model = estim_class_model(features, nb_classes)
    proba = predict_proba(model, features)          # (scikit-learn's arithmetic without its per-call validation: same bits)
    logging.debug('list of probabilities: %r', proba.shape)

    segm = None
    if fused:
        try:
            # fused: unary / edge terms ('model' edges), alpha-expansion and the gather in one call on the prepared graph
            from pyimsegm_amd.graph_cuts import compute_pairwise_cost
            use_gc = not (np.isscalar(gc_regul) and gc_regul <= 0)
            segm_ready()
            segm = sess.segment(compute_pairwise_cost(gc_regul, proba.shape), 'model', proba=proba, use_graphcut=use_gc,
                                pinned=False, segm_out=segm_buf)['segm']
        except _hip.HipFusedPathError as ex:            # (IMSEGM_E_FUSED_PATH only: any other error of the call is an error)
            logging.info('volume graph by neighbour tables (%s)', ex)
    if segm is None:
        graph_labels = segment_graph_cut_general(_ShapeOnly(sess.shape), proba, image, features, gc_regul, _session=sess)
        segm_ready()
        segm, _ = sess.gather(graph_labels, segm_out=segm_buf)     # (into the array touched above: no second result array)
    return segm
