"""The class-model variants of graph_cuts.estim_class_model (reference graph_cuts.py:73-193) against vectors the
reference produced (tests/golden/make_golden_class_models.py, conda Python with scikit-image 0.18.3); CPU only.
ADVICE r4: every `estim_model` the pipelines document works on a standalone install."""
import builtins
import os
import warnings

import numpy as np
import pytest

from pyimsegm_amd import graph_cuts

G = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'class_models.npz'))


def test_otsu_rule_without_scikit_image(monkeypatch):
    real_import = builtins.__import__

    def no_skimage(name, *a, **k):
        if name.startswith('skimage'):
            raise ImportError(name)
        return real_import(name, *a, **k)

    monkeypatch.setattr(builtins, '__import__', no_skimage)
    for t, want in enumerate(G['otsu_thresholds']):
        assert graph_cuts._threshold_otsu(G['otsu_samples_%d' % t]) == want


def test_multivariate_otsu_equals_the_reference():
    for t in range(4):
        got = graph_cuts.compute_multivarian_otsu(G['features_%d' % t])
        assert got.dtype == bool and np.array_equal(got, G['multivariate_otsu_%d' % t])


@pytest.mark.parametrize('row', range(len(G['plan_names'])))
def test_every_estim_model_fits_with_the_reference_configuration(row):
    assert G['plan_same_as_reference_in_generator'][row] == 1          # (same interpreter, same seed: identical output)
    name, nb = str(G['plan_names'][row]), int(G['plan_classes'][row])
    kw = {'GMM': {}, 'GMM_kmeans': dict(pca_coef=0.95, max_iter=5), 'GMM_Otsu': dict(max_iter=5),
          'kmeans_quantiles': dict(use_scaler=False, max_iter=5), 'kmeans': dict(max_iter=9), 'BGM': dict(max_iter=5),
          'Otsu': dict(max_iter=3)}[name]
    fts = G['plan_features']
    np.random.seed(11)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        model = graph_cuts.estim_class_model(fts, nb, estim_model=name, **kw)
    last = model.steps[-1][1]
    assert type(last).__name__ == str(G['plan_mixture'][row])
    assert (last.n_init, last.max_iter) == (int(G['plan_n_init'][row]), int(G['plan_max_iter'][row]))
    assert [n for n, _ in model.steps] == (['std_scaler'] if kw.get('use_scaler', True) else []) + \
        (['reduce_dim'] if 'pca_coef' in kw else []) + ['model']
    proba = model.predict_proba(fts)
    assert proba.shape == (len(fts), nb) and np.allclose(proba.sum(axis=1), 1)


def test_kmeans_model_and_labels():
    np.random.seed(0)
    fts = np.vstack([np.random.random((50, 3)) - 1, np.random.random((50, 3)) + 1])
    model, labels = graph_cuts.estim_class_model_kmeans(fts, 2, max_iter=9)
    assert labels.shape == (100, ) and model.predict_proba(fts).shape == (100, 2)
    assert len(set(labels[:50])) == 1 and len(set(labels[50:])) == 1 and labels[0] != labels[-1]
    model, labels = graph_cuts.estim_class_model_kmeans(fts, 2, init_type='quantiles')
    assert len(set(labels[:50])) == 1 and labels[0] != labels[-1]


@pytest.mark.parametrize('kind,n_init', [('GaussianMixture', 5), ('BayesianGaussianMixture', 3)])
def test_restarts_side_by_side_give_the_parameters_of_fit(kind, n_init):
    """graph_cuts.fit_mixture_restarts: the restarts of scikit-learn's EM loop in worker threads -- fitted parameters, lower bound,
    iteration count and the state of the random stream afterwards bit for bit those of `mixture.fit` (120 000 x 3: above the size
    where a multi-threaded BLAS splits its reductions, the case that needs the one-thread pools inside)"""
    from sklearn import mixture
    rng = np.random.default_rng(7)
    n = 120000
    table = np.vstack([rng.normal([0, 0, 0], [1, .5, .8], (n // 3, 3)), rng.normal([3, 1, 2], [.7, .6, .5], (n // 3, 3)),
                       rng.normal([-2, 2, 1], [.5, .9, .6], (n - 2 * (n // 3), 3))]) + rng.normal(0, 2., (n, 3))
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        np.random.seed(5)
        plain = getattr(mixture, kind)(n_components=3, covariance_type='full', n_init=n_init, max_iter=30).fit(table)
        after_plain = np.random.random()
        np.random.seed(5)
        ours = graph_cuts.fit_mixture_restarts(getattr(mixture, kind)(n_components=3, covariance_type='full', n_init=n_init, max_iter=30),
                                               table, workers=4)
        after_ours = np.random.random()
    for name in ('weights_', 'means_', 'covariances_', 'precisions_cholesky_', 'precisions_'):
        assert np.array_equal(getattr(plain, name), getattr(ours, name)), name
    assert (plain.n_iter_, plain.lower_bound_, plain.converged_) == (ours.n_iter_, ours.lower_bound_, ours.converged_)
    assert after_plain == after_ours
    assert np.array_equal(plain.predict_proba(table[:2000]), ours.predict_proba(table[:2000]))


def test_restarts_fall_back_to_fit(monkeypatch):
    """one worker, or something unexpected inside the side-by-side path (here: the thread-pool limiter refuses): scikit-learn's own
    loop from where the random stream stood -- same result as `fit`"""
    import threadpoolctl
    from sklearn import mixture
    rng = np.random.default_rng(1)
    table = np.vstack([rng.normal(0, 1, (300, 2)), rng.normal(4, 1, (300, 2))])
    monkeypatch.setattr(graph_cuts, '_RESTARTS_SIDE_BY_SIDE_FROM', 2)            # (small tables take plain `fit` anyway)
    np.random.seed(2)
    want = mixture.GaussianMixture(2, n_init=3).fit(table).means_

    def refuse(*a, **k):
        raise RuntimeError('no limiter')

    for workers, broken in ((1, False), (4, True)):
        monkeypatch.setattr(graph_cuts, '_RESTARTS_SIDE_BY_SIDE_FROM', 2)
        if broken:
            monkeypatch.setattr(threadpoolctl, 'threadpool_limits', refuse)
        np.random.seed(2)
        got = graph_cuts.fit_mixture_restarts(mixture.GaussianMixture(2, n_init=3), table, workers=workers).means_
        monkeypatch.undo()
        assert np.array_equal(got, want)


def test_stream_advance_on_few_rows_only_where_it_is_known_to_be_the_same(monkeypatch):
    """the one-iteration k-means fits that advance the random stream run on 256 rows only for scikit-learn releases whose seeding
    draws the same whatever the number of rows (and after a check on a private stream); for 1.0 - 1.2 (randint(n_samples)) on all"""
    import sklearn
    rng = np.random.default_rng(3)
    table = rng.normal(0, 1, (20000, 3))
    graph_cuts._SEEDING_ON_FEW_ROWS.clear()
    rows = graph_cuts._seeding_rows(table, 3)
    major, minor = (int(v) for v in sklearn.__version__.split('.')[:2])
    assert (rows != slice(None)) == ((major, minor) < (1, 0) or (major, minor) >= (1, 3))
    monkeypatch.setattr(sklearn, '__version__', '1.1.3')
    assert graph_cuts._seeding_rows(table, 3) == slice(None)
    assert graph_cuts._seeding_rows(table[:1000], 3) == slice(None)          # (small tables: nothing to save)


def test_touched_result_array():
    """pipelines._touched_result: a small result is a plain array; a large one comes back with worker threads that touch every page
    (one byte per 4 KB) while the caller goes on -- after the join every page has been written"""
    from pyimsegm_amd import pipelines
    small, join = pipelines._touched_result((8, 16, 16))
    assert small.shape == (8, 16, 16) and small.dtype == np.int32
    join()
    big, join = pipelines._touched_result((17, 1024, 1024))         # 68 MB: above the threshold
    big.reshape(-1).view(np.uint8)[4096 * 5 + 1] = 7                 # (a byte the touching never writes)
    join()
    flat = big.reshape(-1).view(np.uint8)
    assert big.shape == (17, 1024, 1024) and big.flags.c_contiguous
    assert not flat[::4096].any() and flat[4096 * 5 + 1] == 7


def test_float32_features_stay_float32_as_in_pipeline_fit():
    """`Pipeline.fit` keeps a float32 table float32 through StandardScaler and the mixture; estim_class_model used to
    cast to float64 first (parameters then differ in the last bits).  Same class, same parameters, same probabilities as the plain
    scikit-learn pipeline on the float32 table -- and a second side-by-side fit while one is running takes scikit-learn's own loop
    (one holder of the process-wide BLAS limit at a time)"""
    from sklearn import mixture, pipeline, preprocessing
    rng = np.random.default_rng(3)
    fts = np.vstack([rng.normal(0, 1, (300, 3)), rng.normal(3, 1, (300, 3))]).astype(np.float32)
    np.random.seed(4)
    want = pipeline.Pipeline([('std_scaler', preprocessing.StandardScaler()),
                              ('model', mixture.GaussianMixture(n_components=2, covariance_type='full', n_init=9, max_iter=99))]).fit(fts)
    np.random.seed(4)
    got = graph_cuts.estim_class_model(fts, 2)
    assert got.steps[-1][1].means_.dtype == want.steps[-1][1].means_.dtype
    for name in ('weights_', 'means_', 'covariances_', 'precisions_cholesky_'):
        assert np.array_equal(getattr(want.steps[-1][1], name), getattr(got.steps[-1][1], name)), name
    assert np.array_equal(want.predict_proba(fts), got.predict_proba(fts))
    # the lock: side-by-side fits take turns (the BLAS limit they hold is process-wide) -- two at once give what two in a row give
    import threading
    table = np.vstack([rng.normal(0, 1, (20000, 2)), rng.normal(4, 1, (20000, 2))])
    got = {}

    def fit(name, seed_table):
        got[name] = graph_cuts.fit_mixture_restarts(mixture.GaussianMixture(2, n_init=3, random_state=7), seed_table, workers=4).means_
    threads = [threading.Thread(target=fit, args=(name, table)) for name in ('a', 'b', 'c')]        # two at once, the third waits
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    plain = mixture.GaussianMixture(2, n_init=3, random_state=7).fit(table)
    assert all(np.array_equal(got[name], plain.means_) for name in ('a', 'b', 'c'))
    assert graph_cuts._SIDE_BY_SIDE.busy() == 0
