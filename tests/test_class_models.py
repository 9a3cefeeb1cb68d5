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
