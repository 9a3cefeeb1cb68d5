"""Golden vectors for the class-model variants of graph_cuts.estim_class_model (reference graph_cuts.py:73-193), made
by the reference itself under the build container's conda Python 3.9 (scikit-image 0.18.3):

    /opt/conda/bin/python3.9 tests/golden/make_golden_class_models.py

* `otsu_samples_*` / `otsu_thresholds`: samples and their `skimage.filters.threshold_otsu` (pins the histogram rule the
  package falls back to when scikit-image is absent);
* `features_*` / `multivariate_otsu_*`: feature tables and the reference's `compute_multivarian_otsu` labelling;
* `plan_*`: for every `estim_model` name the mixture class and the `n_init` / `max_iter` the reference's fitted pipeline
  ends up with (the fitted numbers themselves depend on the scikit-learn version and are compared in THIS script, same
  interpreter, same seed: reference vs package, printed).
Build container only -- the tests read the .npz.
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _reference_env import ROOT, ReferenceEnv  # noqa: E402

NAMES = [('GMM', 3, {}), ('GMM_kmeans', 3, dict(pca_coef=0.95, max_iter=5)), ('GMM_Otsu', 2, dict(max_iter=5)),
         ('kmeans_quantiles', 3, dict(use_scaler=False, max_iter=5)), ('kmeans', 3, dict(max_iter=9)),
         ('BGM', 3, dict(max_iter=5)), ('Otsu', 2, dict(max_iter=3)), ('Otsu', 3, dict(max_iter=3))]


def main():
    out = {}
    rng = np.random.RandomState(5)
    from skimage.filters import threshold_otsu
    thresholds = []
    for t in range(12):
        v = np.concatenate([rng.normal(0, 1, rng.randint(5, 400)),
                            rng.normal(rng.uniform(0, 6), rng.uniform(.2, 2), rng.randint(3, 300))])
        if t % 3 == 0:
            v = np.round(v, 1)
        out['otsu_samples_%d' % t] = v
        thresholds.append(threshold_otsu(v))
    out['otsu_thresholds'] = np.array(thresholds)
    with ReferenceEnv() as env:
        ref = env.graph_cuts
        sys.path.insert(0, ROOT)
        from pyimsegm_amd import graph_cuts as mine
        for t in range(4):
            fts = np.vstack([rng.random_sample((30 + 7 * t, 2 + t)) - 1, rng.random_sample((25, 2 + t)) + rng.uniform(0, 1.5)])
            fts[:, 1] = -fts[:, 1]
            out['features_%d' % t] = fts
            out['multivariate_otsu_%d' % t] = ref.compute_multivarian_otsu(fts)
        fts = np.vstack([rng.random_sample((60, 4)) - 1, rng.random_sample((50, 4)) + 1, rng.random_sample((40, 4)) * 3])
        out['plan_features'] = fts
        rows = []
        for name, nb, kw in NAMES:
            np.random.seed(11)
            a = ref.estim_class_model(fts, nb, estim_model=name, **kw)
            np.random.seed(11)
            b = mine.estim_class_model(fts, nb, estim_model=name, **kw)
            same = np.array_equal(a.predict_proba(fts), b.predict_proba(fts))
            last = a.steps[-1][1]
            rows.append((name, nb, type(last).__name__, last.n_init, last.max_iter, int(same)))
            print(rows[-1])
        out['plan_names'] = np.array([r[0] for r in rows])
        out['plan_classes'] = np.array([r[1] for r in rows])
        out['plan_mixture'] = np.array([r[2] for r in rows])
        out['plan_n_init'] = np.array([r[3] for r in rows])
        out['plan_max_iter'] = np.array([r[4] for r in rows])
        out['plan_same_as_reference_in_generator'] = np.array([r[5] for r in rows])
    np.savez_compressed(os.path.join(ROOT, 'tests', 'golden', 'class_models.npz'), **out)


if __name__ == '__main__':
    main()
