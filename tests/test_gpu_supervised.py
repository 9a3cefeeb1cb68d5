"""Supervised path: superpixel labels from an annotation, classifier training and
segmentation with the trained classifier -- reference imsegm/pipelines.py:272-379 + :160-241."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

FEATS = {'color': ('mean', 'std', 'energy')}


def _annotated(seed, shape=(210, 280)):
    from pyimsegm_amd.utilities.synthetic import voronoi_image
    return voronoi_image(shape[0], shape[1], seed=seed, nb_seeds=9, return_classes=True)


def test_superpixel_labels_from_annotation_match_oracle(oracle):
    from pyimsegm_amd import pipelines as P
    from pyimsegm_amd.utilities import ImageDimensionError
    img, annot = _annotated(3)
    annot = annot.copy()
    annot[:40, :50] = -1                                        # a "do not care" corner
    slic, fts, labels = P.wrapper_compute_color2d_slic_features_labels((img, annot), 16, 0.2, FEATS, 0.9)
    ref_slic = oracle.segment_slic_img2d(img, 16, 0.2)
    assert np.array_equal(slic, ref_slic)
    ann = annot.copy()
    ann[ann < 0] = 3
    hist = oracle.histogram_regions_labels_norm(ref_slic, ann)
    ref = np.argmax(hist, axis=1)
    ref[ref == 3] = -1
    ref[np.max(hist, axis=1) < 0.9] = -1
    assert labels.shape == (ref_slic.max() + 1, ) and np.array_equal(labels, ref)
    assert (labels == -1).any() and set(np.unique(labels)) <= {-1, 0, 1, 2}
    assert fts.shape == (ref_slic.max() + 1, 9) and not np.isnan(fts).any()
    assert annot.min() == -1, 'the caller\'s annotation must not be modified'
    with pytest.raises(ImageDimensionError):
        P.wrapper_compute_color2d_slic_features_labels((img, annot[:, :-1]), 16, 0.2, FEATS, 0.9)


def test_train_classifier_and_segment(oracle):
    from pyimsegm_amd import pipelines as P
    from test_gpu_api import _oracle_pipeline
    pairs = [_annotated(seed) for seed in (11, 12, 13)]
    images, annots = [p[0] for p in pairs], [p[1] for p in pairs]
    np.random.seed(0)
    classif, list_slic, list_fts, list_lbs = P.train_classif_color2d_slic_features(
        images, annots, FEATS, sp_size=16, sp_regul=0.2, clf_name='RandForest', label_purity=0.9,
        feature_balance='unique', nb_workers=2)
    assert len(list_slic) == len(list_fts) == len(list_lbs) == 3
    for slic, fts, lbs, img in zip(list_slic, list_fts, list_lbs, images):
        assert np.array_equal(slic, oracle.segment_slic_img2d(img, 16, 0.2))
        assert len(fts) == len(lbs) == slic.max() + 1
    assert list(classif.classes_) == [0, 1, 2]
    # a new image of the same kind: segmentation with the trained classifier, against the annotation ...
    test_img, test_annot = _annotated(21)
    segm, soft = P.segment_color2d_slic_features_model_graphcut(test_img, classif, FEATS, sp_size=16, sp_regul=0.2,
                                                                gc_regul=1., gc_edge_type='model')
    assert segm.shape == test_annot.shape and soft.shape == test_annot.shape + (3, )
    assert np.mean(segm == test_annot) > 0.95
    # ... and bit for bit against the oracle's stage chain with the same classifier
    _, _, segm_ref, soft_ref = _oracle_pipeline(oracle, test_img, classif, 16, 0.2, 1.)
    assert np.array_equal(segm, segm_ref)
    assert np.allclose(soft, soft_ref, rtol=1e-9, atol=1e-12)
    with pytest.raises(ValueError):
        P.train_classif_color2d_slic_features(images, annots[:-1], FEATS)
