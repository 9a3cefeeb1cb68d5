"""What the supervised pipeline (``pipelines.train_classif_color2d_slic_features``, reference ``pipelines.py:292-379``)
needs from the reference's classifier toolbox ``imsegm/classification.py`` -- and nothing else of it.

The toolbox (a dozen classifiers, feature selection, ROC / export helpers, 1 700 lines) is scikit-learn glue that never
touches the device; SURVEY marks it out of scope.  Three of its names sit on the path between the device-computed superpixel
features and the trained model, so they exist here, written against scikit-learn directly:

* :func:`convert_set_features_labels_2_dataset` -- one training set out of the per-image features / labels
  (``classification.py:1416-1476``);
* :class:`CrossValidateGroups` -- folds that hold out whole images (``classification.py:1616-1700``);
* :func:`create_classif_search_train_export` -- scaler -> (PCA) -> classifier, trained once or after a search
  (``classification.py:656-759``).

Every other name of the reference module resolves to the reference's own function when a reference package is installed
behind the ``imsegm`` overlay (module ``__getattr__``).
"""
import collections
import logging
import random

import numpy as np

from pyimsegm_amd.utilities import reference_attribute

#: classifier the pipelines train when none is named (``classification.py:47``)
DEFAULT_CLASSIF_NAME = 'RandForest'
#: the 'unique' balancing compares feature vectors after rounding to this many digits (``classification.py:55``)
ROUND_UNIQUE_FTS_DIGITS = 3


def __getattr__(name):
    """names of the reference toolbox that are not part of the path: the reference's own, when one is installed"""
    return reference_attribute('classification', name)


def _classifier(name, nb_workers):
    """the classifiers of ``classification.py:98-123`` by name, with the parameters the reference gives them"""
    from sklearn import ensemble, linear_model, neighbors, svm, tree
    makers = {
        'RandForest': lambda: ensemble.RandomForestClassifier(n_estimators=20, min_samples_leaf=2, min_samples_split=3, n_jobs=nb_workers),
        'GradBoost': lambda: ensemble.GradientBoostingClassifier(subsample=0.25, warm_start=False, max_depth=6, min_samples_leaf=6,
                                                                 n_estimators=200, min_samples_split=7),
        'LogistRegr': lambda: linear_model.LogisticRegression(solver='sag', n_jobs=nb_workers),
        'KNN': lambda: neighbors.KNeighborsClassifier(n_jobs=nb_workers),
        'SVM': lambda: svm.SVC(kernel='rbf', probability=True, tol=2e-3, max_iter=5000),
        'DecTree': lambda: tree.DecisionTreeClassifier(),
        'AdaBoost': lambda: ensemble.AdaBoostClassifier(n_estimators=5),
    }
    if name not in makers:
        raise KeyError('unknown classifier %r (known: %s)' % (name, ', '.join(sorted(makers))))
    return makers[name]()


#: a compact search space per classifier (lists: they serve the grid and the randomised search alike)
_SEARCH_SPACES = {
    'RandForest': {'classif__n_estimators': [10, 20, 40, 80], 'classif__min_samples_split': [2, 3, 5, 9],
                   'classif__min_samples_leaf': [1, 2, 5], 'classif__criterion': ['gini', 'entropy']},
    'GradBoost': {'classif__n_estimators': [50, 100, 200], 'classif__max_depth': [2, 4, 6], 'classif__learning_rate': [0.03, 0.1, 0.3]},
    'LogistRegr': {'classif__C': [0.01, 0.1, 1., 10., 100.]},
    'KNN': {'classif__n_neighbors': [3, 5, 9, 15], 'classif__weights': ['uniform', 'distance']},
    'SVM': {'classif__C': [0.1, 1., 10., 100.], 'classif__gamma': ['scale', 0.01, 0.1, 1.]},
    'DecTree': {'classif__max_depth': [None, 4, 8, 16], 'classif__min_samples_leaf': [1, 2, 5]},
    'AdaBoost': {'classif__n_estimators': [5, 15, 50], 'classif__learning_rate': [0.1, 0.5, 1.]},
}


def create_classif_search_train_export(clf_name, features, labels, cross_val=10, nb_search_iter=100, search_type='random',
                                       eval_metric='f1', nb_workers=1, path_out=None, params=None, pca_coef=0.98,
                                       feature_names=None, label_names=None):
    """ ``Pipeline([StandardScaler, (PCA,) classifier])`` trained on the given samples -- at once, or (``nb_search_iter`` > 1
    or a grid search) refitted with the best parameters of a cross-validated search.  Signature and errors of
    ``classification.py:656-759``; the export to ``path_out`` is not part of this path (returned untouched).

    :return tuple(obj,str): trained pipeline, path_out
    """
    from sklearn import decomposition, metrics, model_selection, pipeline, preprocessing
    if not list(labels):
        raise RuntimeError('some labels has to be given')
    samples = np.nan_to_num(features)
    if len(samples) != len(labels):
        raise ValueError('features (%i) and labels (%i) should have equal length' % (len(samples), len(labels)))
    if samples.ndim != 2 or samples.shape[1] < 1:
        raise ValueError('at least one feature is required')
    stages = [('scaler', preprocessing.StandardScaler())]
    if pca_coef is not None:
        stages += [('reduce_dim', decomposition.PCA(pca_coef))]
    model = pipeline.Pipeline(stages + [('classif', _classifier(clf_name, -1))])
    if nb_search_iter > 1 or search_type == 'grid':
        space = _SEARCH_SPACES.get(clf_name, {})
        classes, dense = np.unique(labels, return_inverse=True)                 # the search sees labels 0 .. n-1
        score_fn = {'f1': metrics.f1_score, 'accuracy': metrics.accuracy_score, 'precision': metrics.precision_score,
                    'recall': metrics.recall_score}[eval_metric.lower()]
        scoring = metrics.make_scorer(score_fn, average='weighted' if len(classes) > 2 else 'binary')
        common = dict(scoring=scoring, cv=cross_val, n_jobs=nb_workers, refit=True)
        if search_type == 'grid':
            search = model_selection.GridSearchCV(model, space, **common)
        else:
            combinations = int(np.prod([len(v) for v in space.values()])) if space else 1
            search = model_selection.RandomizedSearchCV(model, space, n_iter=max(1, min(int(nb_search_iter), combinations)), **common)
        search.fit(samples, dense.tolist())
        logging.info('Best score: %r', search.best_score_)
        model = search.best_estimator_
    model.fit(samples, labels)            # with or without a search: (re)trained on the labels as given
    return model, path_out


def _balanced(samples, labels, how):
    """the samples of ONE image thinned per label (``classification.py:1344-1413``): 'random' / 'kmeans' down to the size of the
    rarest label, 'unique' to the distinct vectors after rounding; labels in ascending order, an unknown method only warns"""
    from sklearn import cluster
    samples, labels = np.array(samples), np.asarray(labels)
    target = min(collections.Counter(labels.tolist()).values())
    kind = how.lower()
    if kind not in ('random', 'kmeans', 'unique'):
        logging.warning('not defined balancing method "%s"', how)
    rows, tags = [], []
    for tag in np.unique(labels):
        block = samples[labels == tag, :]
        if kind == 'unique':
            rounded = np.round(np.asarray(block, dtype=np.float64), ROUND_UNIQUE_FTS_DIGITS)
            block = np.unique(rounded, axis=0).reshape(-1, rounded.shape[1])
        elif kind == 'random' and len(block) > target:
            order = list(range(len(block)))
            random.shuffle(order)
            block = block[order[:target], :]
        elif kind == 'kmeans' and len(block) > target:
            to_centres = cluster.KMeans(n_clusters=target, init='random', n_init=3, max_iter=5).fit_transform(block)
            block = block[np.argmin(to_centres, axis=0), :]        # the sample nearest to every centre
        rows += np.asarray(block).tolist()
        tags += [tag.item()] * len(block)
    return np.array(rows), tags


def convert_set_features_labels_2_dataset(imgs_features, imgs_labels, drop_labels=None, balance_type=None):
    """ the per-image features and labels as one training set: images in sorted key order, ``drop_labels`` removed,
    every image balanced on its own (``classification.py:1416-1476``)

    :return tuple(ndarray,ndarray,list(int)): features, labels, number of samples per image
    """
    if any(key not in imgs_labels for key in imgs_features):
        raise ValueError('missing some items of %r' % imgs_labels.keys())
    rows, tags, sizes = [], [], []
    for key in sorted(imgs_features):
        samples, labels = np.array(imgs_features[key]), np.array(imgs_labels[key]).astype(int)
        keep = ~np.isin(labels, list(drop_labels or []))
        samples, labels = samples[keep], labels[keep]
        if balance_type is not None and len(labels) > 0:
            samples, labels = _balanced(samples, labels, balance_type)
        rows += np.asarray(samples).tolist()
        tags += np.asarray(labels).tolist()
        sizes.append(len(labels))
    return np.array(rows), np.array(tags, dtype=int), sizes


class CrossValidateGroups(object):
    """ cross-validation folds over whole sets (images): a fold tests on ``nb_hold_out`` consecutive sets -- a share of
    them when below one -- and trains on all others (``classification.py:1616-1700``)
    """

    def __init__(self, set_sizes, nb_hold_out, rand_seed=None):
        count = len(set_sizes)
        if nb_hold_out <= 0:
            raise ValueError('Number of holdout has to be positive number.')
        if count <= nb_hold_out:
            raise ValueError('Number of holdout has to be smaller then total size.')
        self._per_fold = max(1, int(np.round(count * nb_hold_out)) if nb_hold_out < 1 else int(nb_hold_out))
        ends = np.cumsum([int(size) for size in set_sizes])
        self.set_indexes = [list(range(end - int(size), end)) for end, size in zip(ends, set_sizes)]
        self._order = list(range(count))
        if rand_seed is not None and rand_seed is not False:
            np.random.RandomState(rand_seed).shuffle(self._order)

    def __len__(self):
        return -(-len(self._order) // self._per_fold)

    def _samples(self, sets):
        return [index for which in sorted(sets) for index in self.set_indexes[which]]

    def __iter__(self):
        for first in range(0, len(self._order), self._per_fold):
            held_out = self._order[first:first + self._per_fold]
            yield self._samples(set(self._order) - set(held_out)), self._samples(held_out)
