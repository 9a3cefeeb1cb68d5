"""GraphCut stage of the hot path: class model evaluation, unary / pairwise / edge terms, alpha-expansion.

This module carries the names of the reference's ``imsegm/graph_cuts.py`` that the SLIC -> descriptors -> GraphCut path
touches (same names, argument meaning, error types and messages -- the drop-in contract), written against this repo's
device sessions:

* the region adjacency graph, the superpixel centres and the cut itself (``gco.cut_general_graph`` in the reference,
  ``graph_cuts.py:735-744``) are calls into ``libimsegm_hip.so``;
* the per-edge formulas of ``graph_cuts.py:303-657`` are evaluated on the device inside the fused pipeline call
  (``csrc/terms.hip``); the numpy forms below serve callers that enter at this stage with their own arrays, and the
  parity tests, and give the same integer energies;
* the mixture fit is scikit-learn on the host, configured as ``graph_cuts.py:73-163`` configures it.

The alternative class models of ``graph_cuts.py:73-163`` (``GMM_kmeans``, ``GMM_Otsu``, ``kmeans``, ``kmeans_quantiles``,
``BGM``, ``Otsu``) are configured here too (:func:`estim_class_model`, one table row each): they are public parameters of the
pipelines and must work on a standalone install.  What the path never touches -- ``estim_gmm_params``, the transition-count
helpers, the pairwise-matrix estimators ... -- is NOT restated: with a reference package installed behind the ``imsegm``
overlay those names resolve to the reference's own functions (module ``__getattr__`` below), without one they raise an
``AttributeError`` that says so.
"""
import logging

import numpy as np
from sklearn.utils.extmath import row_norms

from pyimsegm_amd import _hip
from pyimsegm_amd.utilities import reference_attribute

#: number of iterations of the reference's fixed-iteration GraphCut calls (``graph_cuts.py:24``)
DEFAULT_GC_ITERATIONS = 25
#: class probabilities are clipped to [MIN_UNARY_PROB, 1 - MIN_UNARY_PROB] before the logarithm (``graph_cuts.py:26``)
MIN_UNARY_PROB = 0.01
#: ceiling of the pairwise (smoothness) costs (``graph_cuts.py:28``)
MAX_PAIRWISE_COST = 1e5
#: edge weights live in [1 / MIN_MAX_EDGE_WEIGHT, MIN_MAX_EDGE_WEIGHT] (``graph_cuts.py:30``)
MIN_MAX_EDGE_WEIGHT = 1e3

def __getattr__(name):
    """names of the reference module this file does not restate: the reference's own, when one is installed"""
    return reference_attribute('graph_cuts', name)


def cut_grid_graph(unary_cost, pairwise_cost, cost_v, cost_h, n_iter=-1, algorithm='expansion', **kwargs):
    """ drop-in for ``gco.cut_grid_graph`` (reference ``region_growing.py:20,248``) on the GPU """
    return _hip.cut_grid_graph(unary_cost, pairwise_cost, cost_v, cost_h, n_iter=n_iter, algorithm=algorithm)


def cut_general_graph(edges, edge_weights, unary_cost, pairwise_cost, n_iter=-1, algorithm='expansion', **kwargs):
    """ drop-in for ``gco.cut_general_graph`` (reference import ``graph_cuts.py:12-15``) on the GPU """
    return _hip.cut_general_graph(edges, edge_weights, unary_cost, pairwise_cost, n_iter=n_iter, algorithm=algorithm)


def _scaler_mixture_steps(model):
    """(scalers, mixture) when ``model`` is ``Pipeline([StandardScaler ...,] GaussianMixture)``, else None"""
    from sklearn.mixture import GaussianMixture
    from sklearn.pipeline import Pipeline
    from sklearn.preprocessing import StandardScaler
    if not isinstance(model, Pipeline):
        return None
    *front, (_, last) = model.steps
    if type(last) is not GaussianMixture or not hasattr(last, '_estimate_weighted_log_prob'):
        return None
    if any(type(step) is not StandardScaler for _, step in front):
        return None
    return [step for _, step in front], last


def predict_proba(model, features):
    """ ``model.predict_proba(features)``; for the model the pipelines fit (scaler + mixture) without scikit-learn's
    per-call input validation -- the same arithmetic (``StandardScaler.transform``,
    ``GaussianMixture._estimate_log_prob_resp``) bit for bit, a third of the host time of a pipeline step when several
    images are in flight and the worker threads share the interpreter lock """
    try:
        parts = _scaler_mixture_steps(model)
        table = np.array(features, dtype=np.float64)
        if parts is not None and table.ndim == 2 and np.isfinite(table).all():
            from scipy.special import logsumexp
            for scaler in parts[0]:
                if scaler.with_mean:
                    table -= scaler.mean_
                if scaler.with_std:
                    table /= scaler.scale_
            weighted = parts[1]._estimate_weighted_log_prob(table)
            with np.errstate(under='ignore'):
                return np.exp(weighted - logsumexp(weighted, axis=1)[:, np.newaxis])
    except Exception:       # private scikit-learn API moved: the public call below
        pass
    return model.predict_proba(features)


def compute_multivarian_otsu(features):
    """ a two-class labelling of feature vectors from one Otsu threshold per feature (``graph_cuts.py:166-193``): every
    column votes ``value > its threshold``; a column after the first votes the other way round when that agrees better
    with the mean vote of the columns before it; a sample is True when more than half of its columns say so
    """
    table = np.asarray(features)
    votes = np.zeros(table.shape)
    for col in range(table.shape[-1]):
        vote = table[:, col] > _threshold_otsu(table[:, col])
        if col:
            consensus = votes[:, :col].mean(axis=1)
            if np.abs(~vote - consensus).mean() < np.abs(vote - consensus).mean():
                vote = ~vote
        votes[:, col] = vote
    return votes.mean(axis=1) > 0.5


def _threshold_otsu(values, nbins=256):
    """Otsu's threshold of a sample over a 256-bin histogram -- ``skimage.filters.threshold_otsu`` when scikit-image is
    installed, otherwise the same published rule (the bin centre that maximises the between-class variance
    w0 w1 (m0 - m1)^2 of the split behind it)"""
    try:
        from skimage.filters import threshold_otsu
        return threshold_otsu(values, nbins)
    except ImportError:
        pass
    counts, rims = np.histogram(np.ravel(values), nbins)
    mids = (rims[:-1] + rims[1:]) / 2.
    counts = counts.astype(float)
    below, above = np.cumsum(counts), np.cumsum(counts[::-1])[::-1]
    with np.errstate(divide='ignore', invalid='ignore'):
        mean_below = np.cumsum(counts * mids) / below
        mean_above = (np.cumsum((counts * mids)[::-1]) / above[::-1])[::-1]
    between = below[:-1] * above[1:] * (mean_below[:-1] - mean_above[1:])**2
    return mids[:-1][np.argmax(between)]


def estim_class_model_kmeans(features, nb_classes, init_type='k-means++', max_iter=99):
    """ k-means labels and a one-step mixture on the same features (``graph_cuts.py:255-285``): ``'quantiles'`` starts two
    Lloyd iterations from the 5 .. 95 % percentiles of every feature, anything else is scikit-learn's initialisation of
    that name with int(sqrt(max_iter)) restarts.  Returns (mixture, labels).
    """
    from sklearn import cluster, mixture
    if init_type == 'quantiles':
        start = np.percentile(features, np.linspace(5, 95, nb_classes).tolist(), axis=0)
        kmeans = cluster.KMeans(nb_classes, init=np.array(start), max_iter=2)
    else:
        kmeans = cluster.KMeans(nb_classes, init=init_type, max_iter=max_iter, n_init=max(1, int(np.sqrt(max_iter))))
    labels = kmeans.fit_predict(features)
    return mixture.GaussianMixture(nb_classes, covariance_type='full', max_iter=1).fit(features, labels), labels


def _class_model_plan(estim_model, nb_classes):
    """(mixture class name, parameter overrides, labelling run before the fit) for an ``estim_model`` of
    ``graph_cuts.py:73-163``.  The labelling does not steer the fit (scikit-learn's mixtures ignore ``y``); it runs because
    the reference runs it: the k-means variants draw from numpy's global random stream, which the mixture's own
    initialisation continues."""
    parts = estim_model.split('_')
    family, start = parts[0], (parts[-1] if len(parts) > 1 else '')
    from sklearn import cluster
    if family == 'GMM' and start == 'kmeans':
        return 'GaussianMixture', dict(n_init=1), lambda fts, it: cluster.KMeans(nb_classes, init='k-means++').fit_predict(fts)
    if family == 'GMM' and start == 'Otsu':
        return 'GaussianMixture', dict(n_init=1), lambda fts, it: compute_multivarian_otsu(fts)
    if family == 'kmeans':
        how = 'quantiles' if start == 'quantiles' else 'k-means++'
        return 'GaussianMixture', dict(max_iter=1), lambda fts, it: estim_class_model_kmeans(fts, nb_classes, how, it)[1]
    if family == 'BGM':
        return 'BayesianGaussianMixture', {}, None
    if family == 'Otsu' and nb_classes == 2:
        return 'GaussianMixture', dict(max_iter=1, n_init=1), lambda fts, it: compute_multivarian_otsu(fts)
    return 'GaussianMixture', {}, None           # 'GMM', and -- as in the reference -- any name it does not know


def _fit_workers():
    """threads the restarts of a mixture fit may use: the CPUs this process has been placed on (a rank of a multi-GPU run is
    bound to the cores next to its GPU, ``distributed.Group``); ``IMSEGM_FIT_WORKERS`` overrides (1: scikit-learn's own loop)"""
    import os
    env = os.environ.get('IMSEGM_FIT_WORKERS', '')
    if env.isdigit():
        return int(env)
    try:
        return len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        return os.cpu_count() or 1


def _advance_like_kmeans(stream, table, n_clusters):
    """draw from ``stream`` what ``KMeans(n_clusters, n_init=1, random_state=stream).fit`` draws: a fit of one iteration"""
    import warnings
    from sklearn import cluster
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        cluster.KMeans(n_clusters=n_clusters, n_init=1, max_iter=1, random_state=stream).fit(table)


_SEEDING_ON_FEW_ROWS = {}


def _seeding_rows(table, n_clusters):
    """rows of ``table`` on which a one-iteration k-means advances a random stream exactly as on the whole table.

    What ``KMeans.fit`` draws per initialisation does not depend on the rows in the scikit-learn versions this package meets
    (0.24: one integer seed per initialisation; 1.3 and later: one uniform for the first centre and n_local_trials uniforms per
    further centre) -- then 256 rows do, and the nine fits that only exist to advance the stream cost a millisecond each instead
    of fifty.  Releases 1.0 - 1.2 drew the first centre with ``randint(n_samples)`` (rejection sampling: the number of raw draws
    depends on n_samples): there, and wherever a two-trial check on a private stream disagrees, all rows are used."""
    import sklearn
    n = len(table)
    few = slice(0, n, max(1, n // 256))
    if n <= 4096:
        return slice(None)
    key = (sklearn.__version__, n, n_clusters)
    if key not in _SEEDING_ON_FEW_ROWS:
        try:
            major, minor = (int(v) for v in sklearn.__version__.split('.')[:2])
            trusted = (major, minor) < (1, 0) or (major, minor) >= (1, 3)
            private = np.random.RandomState(20260926)
            for _ in range(2 if trusted else 0):
                before = private.get_state()
                _advance_like_kmeans(private, table, n_clusters)
                whole = private.get_state()
                private.set_state(before)
                _advance_like_kmeans(private, table[few], n_clusters)
                part = private.get_state()
                trusted = trusted and whole[2:] == part[2:] and np.array_equal(whole[1], part[1])
        except Exception:
            trusted = False
        _SEEDING_ON_FEW_ROWS[key] = trusted
    return few if _SEEDING_ON_FEW_ROWS[key] else slice(None)


class _SideBySideGate(object):
    """Who may run a side-by-side fit right now.  The fits hold the process's BLAS pools at ONE thread -- ``threadpool_limits`` is
    process-wide, so the limit is taken by the first fit that enters and given back by the last one that leaves -- and every
    OpenMP thread of a restart's k-means calls into that BLAS, which keeps per-thread buffers for 64 callers and dies beyond
    (round 5, on the GPU box).  Hence: at most ``slots`` fits at once (volumes in flight from several worker threads; a third
    waits its turn), each with a budget of ``callers`` BLAS callers -- 2 x 27 stays below the 64.  Concurrent callers that leave
    ``random_state`` at None share numpy's global stream, as concurrent callers of scikit-learn itself do: their fits are not
    reproducible run to run."""

    def __init__(self, slots=2, callers=28):
        import threading
        self.slots, self.callers = slots, callers
        self._gate = threading.BoundedSemaphore(slots)
        self._lock = threading.Lock()
        self._active, self._limiter = 0, None

    def __enter__(self):
        self._gate.acquire()
        try:
            with self._lock:
                if self._active == 0:
                    from threadpoolctl import threadpool_limits
                    self._limiter = threadpool_limits(limits=1, user_api='blas')
                self._active += 1
        except BaseException:
            self._gate.release()
            raise
        return self

    def __exit__(self, *exc):
        with self._lock:
            self._active -= 1
            if self._active == 0 and self._limiter is not None:
                try:
                    self._limiter.restore_original_limits()
                finally:
                    self._limiter = None
        self._gate.release()
        return False

    def busy(self):
        return self._active


def _gate_from_env():
    """``IMSEGM_FIT_GATE=slots,callers`` (experiments; slots x callers must stay below the 64 BLAS buffers)"""
    import os
    try:
        slots, callers = (int(v) for v in os.environ.get('IMSEGM_FIT_GATE', '').split(','))
        if slots >= 1 and callers >= 10 and slots * callers < 64:
            return _SideBySideGate(slots, callers)
    except ValueError:
        pass
    return _SideBySideGate()


_SIDE_BY_SIDE = _gate_from_env()


#: rows from which the restarts of a mixture fit run side by side (below, the thread pools, the stream bookkeeping and the one-thread
#: BLAS cost more than they save: 2 500 x 3 rows fit in 0.1 s either way -- measured 0.22 s side by side)
_RESTARTS_SIDE_BY_SIDE_FROM = 32768


def fit_mixture_restarts(mixture, table, workers=None):
    """ ``mixture.fit(table)`` with the ``n_init`` restarts of scikit-learn's EM loop (``BaseMixture.fit_predict``) run side by side.

    What ``fit`` does, in its order: every restart draws its initialisation from ONE random stream, one restart after the other;
    the EM iterations that follow use no random numbers and nothing of the other restarts.  Here the restarts run in worker
    threads (numpy releases the interpreter lock), each on its own shallow copy of the estimator and from the state the stream of
    ``fit`` has when that restart begins -- for the default k-means initialisation the states are found by advancing the stream
    with one-iteration k-means fits (only the seeding of ``KMeans.fit`` draws), for the cheap initialisations by running them in
    order --, with the BLAS pools held at one thread (several threads calling a multi-threaded BLAS at once split its long
    reductions differently from run to run); the best restart is then picked by ``fit``'s rule (first largest lower bound).
    Same scikit-learn arithmetic call by call: the fitted parameters are bit for bit those of ``mixture.fit(table)``
    (tests/test_class_models.py), in a fifth of the time at the 298 116 x 3 table of a 64 x 4096 x 4096 volume, where the fit was
    80 % of the pipeline.  Anything unexpected -- a scikit-learn whose private methods have moved, warm starts, a single restart,
    a small table -- falls back to ``mixture.fit`` on the restored random stream.
    """
    import copy
    import warnings
    from concurrent.futures import ThreadPoolExecutor
    from sklearn.exceptions import ConvergenceWarning
    from sklearn.utils import check_random_state
    workers = _fit_workers() if workers is None else workers
    needed = ('_initialize_parameters', '_e_step', '_m_step', '_compute_lower_bound', '_get_parameters', '_set_parameters',
              '_check_parameters')
    table = np.asarray(table)
    if workers < 2 or getattr(mixture, 'n_init', 1) < 2 or mixture.max_iter < 1 or getattr(mixture, 'warm_start', False) \
            or getattr(mixture, 'verbose', 0) or any(not hasattr(mixture, name) for name in needed) \
            or table.ndim != 2 or table.dtype != np.float64 or len(table) < max(_RESTARTS_SIDE_BY_SIDE_FROM, mixture.n_components) \
            or not np.isfinite(table).all():
        return mixture.fit(table)
    stream = check_random_state(mixture.random_state)
    stream_state = stream.get_state()
    gate = _SIDE_BY_SIDE
    try:
        from threadpoolctl import threadpool_limits
        table = np.ascontiguousarray(table)
        if hasattr(mixture, '_check_initial_parameters'):          # scikit-learn < 1.2: the generic checks, then the estimator's
            mixture._check_initial_parameters(table)
        else:
            if hasattr(mixture, '_validate_params'):
                mixture._validate_params()
            mixture._check_parameters(table)
        mixture.n_features_in_ = table.shape[1]

        def expectation_maximisation(start):
            own = copy.copy(mixture)
            own._set_parameters(start)
            bound, bounds, converged, n_iter = -np.inf, [], False, 0
            for n_iter in range(1, own.max_iter + 1):
                before = bound
                log_prob_norm, log_resp = own._e_step(table)
                own._m_step(table, log_resp)
                bound = own._compute_lower_bound(log_resp, log_prob_norm)
                bounds.append(bound)
                if abs(bound - before) < own.tol:
                    converged = True
                    break
            return bound, own._get_parameters(), n_iter, bounds, converged

        def restart_from(state):
            # a restart whose k-means initialisation draws from its own copy of the stream, set to where the stream of `fit`
            # stands when that restart begins
            own = copy.copy(mixture)
            private = np.random.RandomState()
            private.set_state(state)
            # (the size of an OpenMP team is a per-thread setting: a limit set in the caller's thread does not reach this one,
            # which would start a team of every core of the host -- nine of them at once is what crashed the process)
            with threadpool_limits(limits=team, user_api='openmp'):
                own._initialize_parameters(table, private)
            return expectation_maximisation(own._get_parameters())

        # Thread budget.  k-means (the initialisation of a restart) runs its Lloyd iterations on OpenMP threads, and EVERY one of
        # them calls into a BLAS: nine restarts side by side on a 256-core host were 9 x 256 callers at once (a worker thread does
        # not inherit the caller's OpenMP limit) -- far more than the 64 an OpenBLAS build keeps per-thread buffers for, and the
        # process died in it (measured on the GPU box, round 5).  The rule here: never more callers at once than ONE k-means of
        # plain scikit-learn would bring in this process (its OpenMP team as it is set right now), and never more than the gate's budget (28: two fits may run at once); the
        # teams of the restarts are cut accordingly, and when that leaves less than one thread per restart the initialisations
        # run one after the other.
        n_workers = max(1, min(workers, mixture.n_init))
        side_by_side_init = getattr(mixture, 'init_params', None) == 'kmeans'
        team = None
        if side_by_side_init:
            from threadpoolctl import threadpool_info
            omp_now = max([int(p.get('num_threads', 1)) for p in threadpool_info() if p.get('user_api') == 'openmp'] or [1])
            team = max(8, min(omp_now, gate.callers)) // n_workers - 1
            if team < 1:
                side_by_side_init, team = False, None
        with gate:
            with ThreadPoolExecutor(max_workers=n_workers) as pool:
                pending = []
                if side_by_side_init:
                    # scikit-learn's default initialisation is most of what is left of the fit (k-means on all rows per
                    # restart), and only its SEEDING draws from the stream (`KMeans.fit`: `_init_centroids`; the Lloyd iterations
                    # use no random numbers).  A k-means of ONE iteration on the same rows advances the stream exactly as the
                    # full one does -- whatever this scikit-learn's `KMeans.fit` draws, in its order --, so the states in
                    # front of the restarts are known after nine cheap fits and the restarts run side by side from their start.
                    seeding_table = table[_seeding_rows(table, mixture.n_components)]
                    for _ in range(mixture.n_init):
                        pending.append(pool.submit(restart_from, stream.get_state()))
                        _advance_like_kmeans(stream, seeding_table, mixture.n_components)
                else:
                    for _ in range(mixture.n_init):
                        mixture._initialize_parameters(table, stream)
                        pending.append(pool.submit(expectation_maximisation, mixture._get_parameters()))
                runs = [job.result() for job in pending]
    except Exception as ex:     # private scikit-learn API moved: its own loop, from where the stream stood
        logging.debug('mixture restarts side by side not available (%r): scikit-learn\'s own loop', ex)
        stream.set_state(stream_state)
        return mixture.fit(table)
    best, best_bound = None, -np.inf
    for run in runs:
        if run[0] > best_bound or best_bound == -np.inf:
            best_bound, best = run[0], run
    mixture._set_parameters(best[1])
    mixture.n_iter_, mixture.lower_bound_ = best[2], best_bound
    # `converged_` and `lower_bounds_` as THIS scikit-learn's `fit` leaves them: newer releases keep the flag of the best restart and
    # the bounds of its iterations, older ones raise one flag when any restart converges and have no `lower_bounds_`
    try:
        import inspect
        from sklearn.mixture._base import BaseMixture
        source = inspect.getsource(BaseMixture.fit_predict)
        per_restart, keeps_bounds = 'self.converged_ = converged' in source, 'lower_bounds_' in source
    except Exception:
        per_restart, keeps_bounds = True, hasattr(mixture, 'lower_bounds_')
    mixture.converged_ = bool(best[4]) if per_restart else any(bool(run[4]) for run in runs)
    if keeps_bounds:
        mixture.lower_bounds_ = best[3]
    if not mixture.converged_:
        warnings.warn('Best performing initialization did not converge. Try different init parameters, or increase max_iter, '
                      'tol, or check for degenerate data.', ConvergenceWarning)
    return mixture


def estim_class_model(features, nb_classes, estim_model='GMM', pca_coef=None, use_scaler=True, max_iter=99):
    """ the class model of the unsupervised pipelines, fitted on the superpixel features with scikit-learn on the host:
    ``Pipeline([StandardScaler,] [PCA,] mixture(full covariance, int(sqrt(max_iter)) restarts))`` -- what
    ``graph_cuts.py:73-163`` builds.  ``estim_model='GMM'`` is the model the hot path uses and the device evaluates; the
    other names of the reference ('GMM_kmeans', 'GMM_Otsu', 'kmeans', 'kmeans_quantiles', 'BGM', 'Otsu') change the
    mixture's parameters as :func:`_class_model_plan` lists them.
    """
    from sklearn import decomposition, mixture, pipeline, preprocessing
    kind, overrides, labelling = _class_model_plan(estim_model, nb_classes)
    params = dict(n_components=nb_classes, covariance_type='full', n_init=max(1, int(np.sqrt(max_iter))), max_iter=max_iter)
    params.update(overrides)
    steps = []
    if use_scaler:
        steps += [('std_scaler', preprocessing.StandardScaler())]
    if pca_coef is not None:
        steps += [('reduce_dim', decomposition.PCA(pca_coef))]
    steps += [('model', getattr(mixture, kind)(**params))]
    if labelling is not None:
        labelling(features, max_iter)       # (scikit-learn's mixtures ignore `y`; the labelling runs for the random stream, see the plan)
    model = pipeline.Pipeline(steps)
    table = np.asarray(features)
    if table.dtype not in (np.float32, np.float64):       # (Pipeline.fit keeps a float32 table float32 through scaler and mixture)
        table = table.astype(np.float64)
    for _, step in model.steps[:-1]:        # what Pipeline.fit does with the steps in front of the last one
        table = step.fit_transform(table)
    fit_mixture_restarts(model.steps[-1][1], table)
    return model


def get_vertexes_edges(segments):
    """ (vertices, edges) of the region adjacency graph of a 2D / 3D label map (``graph_cuts.py:276-300``) """
    from pyimsegm_amd import superpixels
    segments = np.asarray(segments)
    build = {2: superpixels.make_graph_segm_connect_grid2d_conn4, 3: superpixels.make_graph_segm_connect_grid3d_conn6}
    return build[segments.ndim](segments) if segments.ndim in build else (None, None)


def _dense_centres(centres):
    """K x ndim float64 table from whatever ``superpixel_centers`` style input (missing entries: zeros, as
    ``np.nan_to_num`` of the reference's NaN rows)"""
    if isinstance(centres, np.ndarray) and centres.ndim == 2 and centres.dtype == np.float64:
        return np.nan_to_num(centres)                  # straight from the device
    rows = [tuple(c) if c is not None and len(c) else None for c in centres]
    ndim = max(len(r) for r in rows if r is not None)
    return np.nan_to_num(np.array([r if r is not None else (np.nan, ) * ndim for r in rows], dtype=np.float64))


def compute_spatial_dist(centres, edges, relative=False):
    """ Euclidean distance between the centres of connected superpixels (``graph_cuts.py:303-336``)
    """
    pairs = np.asarray(edges)
    if np.max(pairs) >= len(centres):
        raise ValueError('max vertex %i exceed size of centres %i' % (np.max(edges), len(centres)))
    table = _dense_centres(centres)
    # (what sklearn's paired_euclidean_distances evaluates, bit for bit, without its input validation)
    dist = row_norms(table[pairs[:, 0]] - table[pairs[:, 1]])
    return dist / np.mean(dist) if relative else dist


def _similarity(dist):
    """``exp(-d / (2 std(d)^2))`` with the standard deviation of the distance vector itself, as the reference has it
    (``graph_cuts.py:423-435``; equal distances give the NaN / inf numpy gives there)"""
    dist = np.asarray(dist, dtype=float)
    return np.exp(-(dist / (2 * np.std(dist)**2)))


def compute_edge_model(edges, proba, metric='l_T'):
    """ edge weights from the class probabilities of the two end superpixels (``graph_cuts.py:383-439``): an l1, l2 or
    largest-squared-difference ('lT') distance through ``exp(-dist / (2 std(dist)^2))``
    """
    pairs = np.asarray(edges)
    if np.max(pairs) >= len(proba):
        raise ValueError('max vertex %i exceed size of proba %r' % (np.max(edges), proba.shape))
    delta = proba[pairs[:, 0]] - proba[pairs[:, 1]]
    distances = {'l1': lambda: np.abs(delta).sum(axis=-1),             # == paired_manhattan_distances
                 'l2': lambda: row_norms(delta),                        # == paired_euclidean_distances
                 'lT': lambda: np.max(delta**2, axis=1)}
    if metric not in distances:
        logging.error('not implemented for: %s', metric)
        return np.ones(len(pairs))
    dist = distances[metric]()
    return np.exp(-dist / (2 * np.std(dist)**2))


def create_pairwise_matrix(gc_regul, nb_classes):
    """ C x C smoothness matrix from a scalar (uniform, zero diagonal), a list ``[((i, j), weight), ...]`` of symmetric
    entries over a matrix of ones, or a full matrix shifted to a zero minimum (``graph_cuts.py:442-515``)
    """
    if isinstance(gc_regul, np.ndarray):
        if gc_regul.shape != (nb_classes, nb_classes):
            raise ValueError('GC regul matrix %r should match match number of classes (%i)' % (gc_regul.shape, nb_classes))
        return gc_regul - np.min(gc_regul)
    off_diagonal = np.ones(nb_classes) - np.eye(nb_classes)
    if isinstance(gc_regul, list):
        for (i, j), weight in gc_regul:
            off_diagonal[i, j] = off_diagonal[j, i] = weight
        return off_diagonal
    return off_diagonal * gc_regul


def compute_unary_cost(proba, min_prob=MIN_UNARY_PROB):
    """ ``|-log(clip(proba, min_prob, 1 - min_prob))|`` (``graph_cuts.py:518-540``)
    """
    clipped = np.clip(np.array(proba, dtype=np.float64), min_prob, 1 - min_prob)
    return np.abs(-np.log(clipped))


def compute_pairwise_cost(gc_regul, proba_shape, max_pairwise_cost=MAX_PAIRWISE_COST):
    """ the smoothness matrix of :func:`create_pairwise_matrix`, capped (``graph_cuts.py:543-555``) """
    return np.minimum(np.array(create_pairwise_matrix(gc_regul, proba_shape[1]), dtype=np.float64), max_pairwise_cost)


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
    """ store the intermediate results of the cut in ``debug_visual``; with the reference's drawing module installed
    (``imsegm`` overlay) also the rendered images ``drawing.figure_segm_graphcut_debug`` expects
    (``graph_cuts.py:558-571``: ``imgs_unary_cost``, ``img_graph_edges``, ``img_graph_segm``) """
    if debug_visual is None:
        return
    debug_visual.update(segments=segments, edges=edges, edge_weights=edge_weights, unary_cost=unary_cost, graph_labels=graph_labels)
    draw = _reference_drawing()
    if draw is None:
        return
    from pyimsegm_amd.superpixels import superpixel_centers
    label_map = np.asarray(segments)
    debug_visual['imgs_unary_cost'] = draw.draw_graphcut_unary_cost_segments(label_map, unary_cost)
    debug_visual['img_graph_edges'] = draw.draw_graphcut_weighted_edges(label_map, superpixel_centers(label_map), edges, edge_weights,
                                                                        img_bg=debug_visual.get('slic_mean', None))
    debug_visual['img_graph_segm'] = draw.draw_color_labeling(label_map, graph_labels)


def compute_edge_weights(segments, image=None, features=None, proba=None, edge_type='', _session=None):
    """ edges of the superpixel graph and their weights (reference ``graph_cuts.py:574-657``)

    :param ndarray segments: superpixels
    :param ndarray image: input image (``edge_type='color'``)
    :param ndarray features: superpixel features (``edge_type='features'``)
    :param ndarray proba: class probabilities (``edge_type='model[_l1|_l2|_lT]'``)
    :param str edge_type: '', 'const', 'spatial', 'color', 'features', 'model', 'model_<metric>'
    :return tuple(ndarray,ndarray): int32 edges E x 2, float weights E clipped to [1e-3, 1e3]
    """
    from pyimsegm_amd.superpixels import _graph_from_session, _session_for_labels, superpixel_centers
    if _session is None:
        segments = np.asarray(segments)
    if segments.ndim in (2, 3):
        # graph and centres of the resident label map in one device pass
        sess = _session if _session is not None else _session_for_labels(segments)
        try:
            _, edges, centres, _ = _graph_from_session(sess)
        finally:
            if _session is None:
                sess.close()
        edges = np.array(edges, dtype=np.int32).reshape(-1, 2)
    else:
        edges = np.array(get_vertexes_edges(segments)[1], dtype=np.int32)
        centres = superpixel_centers(segments)
    logging.debug('graph edges %r', edges.shape)
    return edges, edge_weights_from_graph(edges, centres, features, proba, edge_type, image=image, segments=segments)


def edge_weights_from_graph(edges, centres, features=None, proba=None, edge_type='', image=None, segments=None):
    """ the weights of given edges (second half of :func:`compute_edge_weights`, reference ``graph_cuts.py:616-657``)

    :param ndarray edges: int32 E x 2
    :param centres: superpixel centres, dense K x ndim table or list of tuples
    :return ndarray: float weights E clipped to [1e-3, 1e3]
    """
    def needs(value, what, error):          # (the errors of graph_cuts.py:620-640)
        if value is None or (what == 'proba' and len(value) == 0):
            raise error('"%s" is required' % what)

    kind, _, metric = edge_type.partition('_')
    if kind.startswith('model'):
        needs(proba, 'proba', ValueError)
        weights = compute_edge_model(edges, proba, edge_type.rsplit('_', 1)[-1] if metric else 'lT')
    elif edge_type == 'color':
        needs(image, 'image', RuntimeError)
        from pyimsegm_amd.descriptors import compute_selected_features_img2d
        unit = np.array(image, dtype=float)
        if unit.max() > 1:
            unit /= 255.
        means, _ = compute_selected_features_img2d(unit, segments, {'color': ['mean']})
        weights = _similarity(np.abs(means[edges[:, 0]] - means[edges[:, 1]]).sum(axis=-1))
    elif edge_type == 'features':
        needs(features, 'features', RuntimeError)
        from sklearn.preprocessing import StandardScaler
        scaled = StandardScaler().fit_transform(features)
        weights = _similarity(row_norms(scaled[edges[:, 0]] - scaled[edges[:, 1]]))
    else:
        weights = np.ones(len(edges))
    weights = np.array(weights, dtype=float)
    if edge_type in ('model', 'features', 'color', 'spatial'):
        # (centres of unused labels are [-1, -1] on the device too, superpixels.py:218)
        weights /= compute_spatial_dist(centres, edges, relative=True)
    low = 1. / MIN_MAX_EDGE_WEIGHT
    weights[weights < low] = low
    weights[weights > MIN_MAX_EDGE_WEIGHT] = MIN_MAX_EDGE_WEIGHT
    return weights


def segment_graph_cut_general(segments, proba, image=None, features=None, gc_regul=1., edge_type='model', edge_cost=1.,
                              debug_visual=None, _session=None):
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
    proba = np.asarray(proba, dtype=np.float64)
    edges, edge_weights = compute_edge_weights(segments, image, features, proba, edge_type, _session=_session)
    edge_weights = edge_weights * edge_cost
    unary = compute_unary_cost(proba)
    if np.isscalar(gc_regul) and gc_regul <= 0:
        # no smoothness term: the cheapest class per superpixel (graph_cuts.py:729-731)
        chosen = np.argmin(unary, axis=-1).astype(np.int32)
    else:
        pairwise = compute_pairwise_cost(gc_regul, proba.shape)
        logging.debug('graph pairwise coefs: \n%r', pairwise)
        chosen = cut_general_graph(edges, edge_weights, unary, pairwise, algorithm='expansion', n_iter=-1)
    insert_gc_debug_images(debug_visual, segments, chosen, unary, edges, edge_weights)
    return chosen


def count_label_transitions_connected_segments(dict_slics, dict_labels, nb_labels=None):
    """ how often the labels a and b meet across an edge of the (device-built) superpixel graphs of a set of images
    (``graph_cuts.py:750-795``); symmetric, an edge between equal labels counts once
    """
    if not nb_labels:
        nb_labels = int(max(np.max(lbs) for lbs in dict_labels.values())) + 1
    counts = np.zeros((nb_labels, nb_labels))
    for name, slic in dict_slics.items():
        labels = np.asarray(dict_labels[name])
        if np.max(slic) + 1 != len(labels):
            raise ValueError('dims are not matching - max slic (%i) and label (%i)' % (np.max(slic), len(labels)))
        ends = labels[np.asarray(get_vertexes_edges(slic)[1])]
        np.add.at(counts, (ends[:, 0], ends[:, 1]), 1)
        np.add.at(counts, (ends[:, 1], ends[:, 0]), 1)
    counts[np.diag_indices(nb_labels)] /= 2
    return counts
