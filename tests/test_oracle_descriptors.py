"""Pin the oracle's descriptor restatement against the reference's golden vectors
(/root/reference/imsegm/descriptors.py doctests) and against the reference's own compiled
features_cython.pyx (oracle/_ref)."""
import numpy as np
import pytest


def _doctest_image():
    # descriptors.py:218-224
    image = np.zeros((2, 10, 3))
    image[:, 2:6, 0] = 1
    image[:, 3:7, 1] = 3
    image[:, 4:9, 2] = 2
    segm = np.array([[0, 0, 0, 0, 0, 1, 1, 1, 1, 1], [0, 0, 0, 0, 0, 1, 1, 1, 1, 1]])
    return image, segm


def test_color2d_mean_doctest(oracle):
    image, segm = _doctest_image()
    # descriptors.py:225-226
    assert np.array_equal(oracle.color2d_mean(image, segm), np.array([[0.6, 1.2, 0.4], [0.2, 1.2, 1.6]]))


def test_color2d_energy_doctest(oracle):
    image, segm = _doctest_image()
    # descriptors.py:253-254
    assert np.array_equal(oracle.color2d_energy(image, segm), np.array([[0.6, 3.6, 0.8], [0.2, 3.6, 3.2]]))


def test_color2d_std_doctest(oracle):
    image, segm = _doctest_image()
    mean = oracle.color2d_mean(image, segm)
    std = np.sqrt(oracle.color2d_variance(image, segm, mean.astype(np.float32)))
    # descriptors.py:282-283 -- including the float32-mean artefact 0.80000003
    expect = np.array([[0.48989794, 1.46969383, 0.80000003], [0.40000001, 1.46969383, 0.80000001]])
    assert np.allclose(std, expect, rtol=0, atol=5e-9)
    assert abs(std[0, 2] - 0.8) > 1e-8


def test_gray3d_doctests(oracle):
    # descriptors.py:470-478, 501-507, 531-537
    image = np.zeros((2, 3, 8))
    image[0, :, 2:6] = 1
    image[1, :, 3:7] = 3
    segm = np.array([[[0, 0, 0, 0, 1, 1, 1, 1]] * 3, [[2, 2, 2, 2, 3, 3, 3, 3]] * 3])
    mean = oracle.gray3d_stat(image, segm, 'mean')
    assert np.allclose(mean, [0.5, 0.5, 0.75, 2.25])
    assert np.allclose(oracle.gray3d_stat(image, segm, 'energy'), [0.5, 0.5, 2.25, 6.75])
    std = np.sqrt(oracle.gray3d_stat(image, segm, 'var', mean))
    assert np.allclose(std, [0.5, 0.5, 1.299038, 1.299038], atol=1e-6)


def test_empty_labels_stay_zero(oracle):
    # features_cython.pyx:76 -- labels without pixels keep 0
    rng = np.random.default_rng(0)
    img = rng.random((6, 7, 3))
    seg = rng.integers(0, 3, (6, 7)) * 2   # labels 0, 2, 4 -> 1 and 3 empty
    mean = oracle.color2d_mean(img, seg)
    assert np.all(mean[[1, 3]] == 0)


@pytest.mark.parametrize('shape,K,dtype', [((40, 31), 7, 'u8'), ((150, 100), 60, 'f'), ((64, 64), 1, 'u8')])
def test_color2d_matches_reference_pyx(oracle, ref_cython, shape, K, dtype):
    """bit-exact agreement with the reference's own native code (oracle/_ref)"""
    if ref_cython is None:
        pytest.skip('oracle/_ref not built (reference tree absent)')
    rng = np.random.default_rng(3)
    if dtype == 'u8':
        img = rng.integers(0, 256, shape + (3,)).astype(np.float32)
    else:
        img = rng.random(shape + (3,)).astype(np.float32)
    seg = rng.integers(0, K, shape).astype(np.int32)
    mean_ref = np.array(ref_cython.computeColorImage2dMean(img, seg))
    energy_ref = np.array(ref_cython.computeColorImage2dEnergy(img, seg))
    var_ref = np.array(ref_cython.computeColorImage2dVariance(img, seg, mean_ref.astype(np.float32)))
    assert np.array_equal(oracle.color2d_mean(img, seg), mean_ref)
    assert np.array_equal(oracle.color2d_energy(img, seg), energy_ref)
    assert np.array_equal(oracle.color2d_variance(img, seg, mean_ref.astype(np.float32)), var_ref)


def test_gray3d_matches_reference_pyx(oracle, ref_cython):
    if ref_cython is None:
        pytest.skip('oracle/_ref not built (reference tree absent)')
    rng = np.random.default_rng(4)
    img = rng.random((5, 20, 17)).astype(np.float32)
    seg = rng.integers(0, 9, (5, 20, 17)).astype(np.int32)
    mean_ref = np.array(ref_cython.computeGrayImage3dMean(img, seg))
    assert np.array_equal(oracle.gray3d_stat(img, seg, 'mean'), mean_ref)
    assert np.array_equal(oracle.gray3d_stat(img, seg, 'energy'), np.array(ref_cython.computeGrayImage3dEnergy(img, seg)))
    m32 = mean_ref.astype(np.float32)
    assert np.array_equal(oracle.gray3d_stat(img, seg, 'var', m32),
                          np.array(ref_cython.computeGrayImage3dVariance(img, seg, m32)))
