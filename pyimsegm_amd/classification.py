"""Classifier side of the supervised SLIC -> features -> classifier -> GraphCut path.

Host-only scikit-learn plumbing, as in the reference (``imsegm/classification.py``): building the training set from
per-image superpixel features and labels (``convert_set_features_labels_2_dataset`` :1219, ``balance_dataset_by_``
:1183 and its down-samplers), the classifier pipeline (``create_classifiers`` :86, ``create_clf_pipeline`` :127) and
training with an optional hyper-parameter search (``create_classif_search_train_export`` :656, without the
export / reporting half).  Nothing here touches the GPU: the device work of this path is the SLIC, descriptor,
label-histogram, graph and graph-cut kernels that ``pipelines`` calls around it.
"""
import collections
import logging
import random

import numpy as np
from sklearn import cluster, decomposition, ensemble, linear_model, metrics, neighbors, pipeline, preprocessing, svm, tree
from sklearn.model_selection import GridSearchCV, RandomizedSearchCV

#: default (recommended) classifier for supervised segmentation
DEFAULT_CLASSIF_NAME = 'RandForest'
#: rounding of features before the 'unique' balancing
ROUND_UNIQUE_FTS_DIGITS = 3
#: metric name -> scikit-learn score function
DICT_SCORING = {
    'f1': metrics.f1_score,
    'accuracy': metrics.accuracy_score,
    'precision': metrics.precision_score,
    'recall': metrics.recall_score,
}


def create_classifiers(nb_workers=-1):
    """ all supported classifiers with the reference's default parameters (``classification.py:98-123``)

    >>> sorted(create_classifiers())
    ['AdaBoost', 'DecTree', 'GradBoost', 'KNN', 'LogistRegr', 'RandForest', 'SVM']
    """
    return {
        'RandForest': ensemble.RandomForestClassifier(n_estimators=20, min_samples_leaf=2, min_samples_split=3,
                                                      n_jobs=nb_workers),
        'GradBoost': ensemble.GradientBoostingClassifier(subsample=0.25, warm_start=False, max_depth=6, min_samples_leaf=6,
                                                         n_estimators=200, min_samples_split=7),
        'LogistRegr': linear_model.LogisticRegression(solver='sag', n_jobs=nb_workers),
        'KNN': neighbors.KNeighborsClassifier(n_jobs=nb_workers),
        'SVM': svm.SVC(kernel='rbf', probability=True, tol=2e-3, max_iter=5000),
        'DecTree': tree.DecisionTreeClassifier(),
        'AdaBoost': ensemble.AdaBoostClassifier(n_estimators=5),
    }


def create_clf_pipeline(name_classif=DEFAULT_CLASSIF_NAME, pca_coef=0.95):
    """ scaler -> (PCA) -> classifier

    >>> [name for name, _ in create_clf_pipeline('DecTree', None).steps]
    ['scaler', 'classif']
    """
    steps = [('scaler', preprocessing.StandardScaler())]
    if pca_coef is not None:
        steps.append(('reduce_dim', decomposition.PCA(pca_coef)))
    steps.append(('classif', create_classifiers()[name_classif]))
    return pipeline.Pipeline(steps)


def create_clf_param_search_distrib(name_classif=DEFAULT_CLASSIF_NAME):
    """ a compact search space per classifier (lists, so it serves the grid and the randomised search)

    >>> sorted(create_clf_param_search_distrib('KNN'))
    ['classif__n_neighbors', 'classif__weights']
    """
    spaces = {
        'RandForest': {'classif__n_estimators': [10, 20, 40, 80], 'classif__min_samples_split': [2, 3, 5, 9],
                       'classif__min_samples_leaf': [1, 2, 5], 'classif__criterion': ['gini', 'entropy']},
        'GradBoost': {'classif__n_estimators': [50, 100, 200], 'classif__max_depth': [2, 4, 6],
                      'classif__learning_rate': [0.03, 0.1, 0.3]},
        'LogistRegr': {'classif__C': [0.01, 0.1, 1., 10., 100.]},
        'KNN': {'classif__n_neighbors': [3, 5, 9, 15], 'classif__weights': ['uniform', 'distance']},
        'SVM': {'classif__C': [0.1, 1., 10., 100.], 'classif__gamma': ['scale', 0.01, 0.1, 1.]},
        'DecTree': {'classif__max_depth': [None, 4, 8, 16], 'classif__min_samples_leaf': [1, 2, 5]},
        'AdaBoost': {'classif__n_estimators': [5, 15, 50], 'classif__learning_rate': [0.1, 0.5, 1.]},
    }
    return spaces.get(name_classif, {})


def relabel_sequential(labels, uq_labels=None):
    """ map labels onto 0..n-1 keeping their order

    >>> relabel_sequential([0, 0, 0, 5, 5, 5, 0, 5])
    [0, 0, 0, 1, 1, 1, 0, 1]
    """
    labels = np.asarray(labels)
    if uq_labels is None:
        uq_labels = np.unique(labels)
    lut = {lb: i for i, lb in enumerate(uq_labels)}
    return [lut[lb] for lb in labels.tolist()]


def create_classif_search_train_export(clf_name, features, labels, cross_val=10, nb_search_iter=100, search_type='random',
                                       eval_metric='f1', nb_workers=1, path_out=None, params=None, pca_coef=0.98,
                                       feature_names=None, label_names=None):
    """ create a classifier and train it once, or after a hyper-parameter search (``nb_search_iter`` > 1 or grid).

    Signature and error behaviour of ``classification.py:656-759``; the export to ``path_out`` (pickle + CSV
    reports) is not part of this path and ``path_out`` is returned untouched.

    :return tuple(obj,str): trained pipeline, path_out

    >>> np.random.seed(0)
    >>> lbs = np.random.randint(0, 3, 150)
    >>> fts = np.random.random((150, 5)) + np.tile(lbs, (5, 1)).T
    >>> clf, _ = create_classif_search_train_export('DecTree', fts, lbs, nb_search_iter=0)
    >>> float(np.mean(clf.predict(fts) == lbs)) > 0.9
    True
    >>> clf, _ = create_classif_search_train_export('KNN', fts, lbs, nb_search_iter=3, cross_val=3)
    >>> clf.predict_proba(fts).shape
    (150, 3)
    """
    if not list(labels):
        raise RuntimeError('some labels has to be given')
    features = np.nan_to_num(features)
    if len(features) != len(labels):
        raise ValueError('features (%i) and labels (%i) should have equal length' % (len(features), len(labels)))
    if not (features.ndim == 2 and features.shape[1] > 0):
        raise ValueError('at least one feature is required')
    logging.debug('training data: %r, labels (%i): %r', features.shape, len(labels), collections.Counter(labels))
    clf_pipeline = create_clf_pipeline(clf_name, pca_coef)
    if nb_search_iter > 1 or search_type == 'grid':
        space = create_clf_param_search_distrib(clf_name)
        nb_labels = len(np.unique(labels))
        scoring = metrics.make_scorer(DICT_SCORING[eval_metric.lower()], average='weighted' if nb_labels > 2 else 'binary')
        if search_type == 'grid':
            search = GridSearchCV(clf_pipeline, space, scoring=scoring, cv=cross_val, n_jobs=nb_workers, refit=True)
        else:
            nb_comb = int(np.prod([len(v) for v in space.values()])) if space else 1
            search = RandomizedSearchCV(clf_pipeline, space, scoring=scoring, cv=cross_val, n_jobs=nb_workers,
                                        n_iter=max(1, min(int(nb_search_iter), nb_comb)), refit=True)
        search.fit(features, relabel_sequential(labels))
        logging.info('Best score: %r', search.best_score_)
        clf_pipeline = search.best_estimator_
    # with or without a search: (re)train on the given labels
    clf_pipeline.fit(features, labels)
    return clf_pipeline, path_out


def shuffle_features_labels(features, labels):
    """ shuffle features and labels together

    >>> np.random.seed(0)
    >>> fts, lbs = shuffle_features_labels(np.arange(12).reshape(6, 2), [0, 0, 1, 1, 2, 2])
    >>> sorted(lbs.tolist()), fts.shape
    ([0, 0, 1, 1, 2, 2], (6, 2))
    """
    if len(features) != len(labels):
        raise ValueError('features (%i) and labels (%i) should have equal length' % (len(features), len(labels)))
    idx = np.random.permutation(len(labels))
    return np.asarray(features)[idx, :], np.asarray(labels)[idx]


def convert_dict_label_features_2_vectors(dict_features):
    """ {label: features[n_label, F]} -> (features[n, F], labels list), in the dictionary's order

    >>> fts, lbs = convert_dict_label_features_2_vectors({0: np.zeros((2, 3)), 4: np.ones((1, 3))})
    >>> fts.shape, lbs
    ((3, 3), [0, 0, 4])
    """
    features, labels = [], []
    for lb in dict_features:
        features += np.asarray(dict_features[lb]).tolist()
        labels += [lb] * len(dict_features[lb])
    return np.array(features), labels


def compose_dict_label_features(features, labels):
    """ (features, labels) -> {label: features of that label}, labels ascending

    >>> d = compose_dict_label_features(np.arange(8).reshape(4, 2), np.array([1, 0, 1, 1]))
    >>> sorted(d), d[1].tolist()
    ([0, 1], [[0, 1], [4, 5], [6, 7]])
    """
    features = np.array(features)
    labels = np.asarray(labels)
    return {lb.item(): features[labels == lb, :] for lb in np.unique(labels)}


def down_sample_dict_features_random(dict_features, nb_samples):
    """ at most ``nb_samples`` randomly chosen samples per label

    >>> random.seed(0)
    >>> d = down_sample_dict_features_random({0: np.zeros((9, 2)), 1: np.ones((3, 2))}, 5)
    >>> d[0].shape, d[1].shape
    ((5, 2), (3, 2))
    """
    out = {}
    for lb, features in dict_features.items():
        features = np.asarray(features)
        if len(features) <= nb_samples:
            out[lb] = features.copy()
            continue
        idx = list(range(len(features)))
        random.shuffle(idx)
        out[lb] = features[idx[:nb_samples], :]
    return out


def down_sample_dict_features_kmean(dict_features, nb_samples):
    """ per label: k-means with ``nb_samples`` clusters, keep the sample nearest to every centre

    >>> np.random.seed(0)
    >>> d = down_sample_dict_features_kmean({0: np.random.random((40, 3)), 1: np.random.random((4, 3))}, 5)
    >>> d[0].shape, d[1].shape
    ((5, 3), (4, 3))
    """
    out = {}
    for lb, features in dict_features.items():
        features = np.asarray(features)
        if len(features) <= nb_samples:
            out[lb] = features.copy()
            continue
        dist = cluster.KMeans(n_clusters=nb_samples, init='random', n_init=3, max_iter=5).fit_transform(features)
        out[lb] = features[np.argmin(dist, axis=0), :]
    return out


def unique_rows(data):
    """ the distinct rows of a matrix (sorted)

    >>> unique_rows(np.array([[1, 2], [0, 3], [1, 2]])).tolist()
    [[0, 3], [1, 2]]
    """
    return np.unique(np.asarray(data), axis=0)


def down_sample_dict_features_unique(dict_features):
    """ per label: the distinct feature vectors after rounding to ``ROUND_UNIQUE_FTS_DIGITS`` digits

    >>> d = down_sample_dict_features_unique({0: np.array([[0.1234, 1.], [0.1233, 1.], [0.2, 1.]])})
    >>> d[0].tolist()
    [[0.123, 1.0], [0.2, 1.0]]
    """
    out = {}
    for lb, features in dict_features.items():
        features = np.round(np.asarray(features, dtype=np.float64), ROUND_UNIQUE_FTS_DIGITS)
        out[lb] = unique_rows(features).reshape(-1, features.shape[1])
    return out


def balance_dataset_by_(features, labels, balance_type='random', min_samples=None):
    """ balance the number of training examples per class: 'random' / 'kmeans' down to the smallest class (or
    ``min_samples``), 'unique' keeps distinct vectors; an unknown method only logs a warning

    >>> np.random.seed(0)
    >>> fts, lbs = balance_dataset_by_(np.random.random((40, 3)), np.array([0] * 30 + [1] * 10), 'kmeans')
    >>> fts.shape, collections.Counter(lbs)[0]
    ((20, 3), 10)
    """
    logging.debug('balance dataset using "%s"', balance_type)
    if not min_samples:
        min_samples = min(collections.Counter(np.asarray(labels).tolist()).values())
    dict_features = compose_dict_label_features(features, labels)
    kind = balance_type.lower()
    if kind == 'random':
        dict_features = down_sample_dict_features_random(dict_features, min_samples)
    elif kind == 'kmeans':
        dict_features = down_sample_dict_features_kmean(dict_features, min_samples)
    elif kind == 'unique':
        dict_features = down_sample_dict_features_unique(dict_features)
    else:
        logging.warning('not defined balancing method "%s"', balance_type)
    return convert_dict_label_features_2_vectors(dict_features)


def convert_set_features_labels_2_dataset(imgs_features, imgs_labels, drop_labels=None, balance_type=None):
    """ concatenate the per-image features and labels (images in sorted key order), dropping ``drop_labels`` and
    balancing every image on its own

    :return tuple(ndarray,ndarray,list(int)): features, labels, number of samples per image

    >>> np.random.seed(0)
    >>> d_fts = {'a': np.random.random((25, 3)), 'b': np.random.random((30, 3))}
    >>> d_lbs = {'a': np.random.randint(0, 2, 25), 'b': np.random.randint(0, 2, 30)}
    >>> fts, lbs, sizes = convert_set_features_labels_2_dataset(d_fts, d_lbs)
    >>> fts.shape, lbs.shape, sizes
    ((55, 3), (55,), [25, 30])
    """
    if not all(k in imgs_labels for k in imgs_features):
        raise ValueError('missing some items of %r' % imgs_labels.keys())
    drop_labels = [] if drop_labels is None else drop_labels
    features_all, labels_all, sizes = [], [], []
    for name in sorted(imgs_features.keys()):
        features = np.array(imgs_features[name])
        labels = np.array(imgs_labels[name]).astype(int)
        for lb in drop_labels:
            features = features[labels != lb]
            labels = labels[labels != lb]
        if balance_type is not None and len(labels) > 0:
            features, labels = balance_dataset_by_(features, labels, balance_type=balance_type)
        features_all += np.asarray(features).tolist()
        labels_all += np.asarray(labels).tolist()
        sizes.append(len(labels))
    return np.array(features_all), np.array(labels_all, dtype=int), sizes


class CrossValidateGroups(object):
    """ cross-validation folds that hold out whole sets (images): every fold tests on ``nb_hold_out`` consecutive
    sets and trains on the rest (role of ``classification.py:1616``)

    >>> cv = CrossValidateGroups([2, 3, 2, 1], nb_hold_out=2)
    >>> len(cv)
    2
    >>> [(train, test) for train, test in cv]
    [([5, 6, 7], [0, 1, 2, 3, 4]), ([0, 1, 2, 3, 4], [5, 6, 7])]
    """

    def __init__(self, set_sizes, nb_hold_out, rand_seed=None):
        nb_sets = len(set_sizes)
        if nb_hold_out <= 0:
            raise ValueError('Number of holdout has to be positive number.')
        if nb_sets <= nb_hold_out:
            raise ValueError('Number of holdout has to be smaller then total size.')
        self._hold = int(np.round(nb_sets * nb_hold_out)) if nb_hold_out < 1 else int(nb_hold_out)
        self._hold = max(self._hold, 1)
        bounds = np.cumsum([0] + [int(s) for s in set_sizes])
        self.set_indexes = [list(range(bounds[i], bounds[i + 1])) for i in range(nb_sets)]
        self._order = list(range(nb_sets))
        if rand_seed is not None and rand_seed is not False:
            np.random.RandomState(rand_seed).shuffle(self._order)

    def __len__(self):
        return int(np.ceil(len(self._order) / float(self._hold)))

    def __iter__(self):
        for i in range(0, len(self._order), self._hold):
            test_sets = self._order[i:i + self._hold]
            train_sets = [s for s in self._order if s not in test_sets]
            yield ([j for s in sorted(train_sets) for j in self.set_indexes[s]],
                   [j for s in sorted(test_sets) for j in self.set_indexes[s]])
