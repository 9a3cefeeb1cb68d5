"""The reference's own run AT THE SIZE of BASELINE configs 3 / 4 / 5 (tests/golden/make_golden_configs.py -> reference_c{3,4,5}.npz)
against the CPU oracle chain: superpixel / supervoxel maps and segmentations by CRC32, descriptors within 1e-5."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


@pytest.fixture(scope='module')
def oracle():
    from oracle import oracle as orc
    orc.lib()
    return orc


def test_config5_volume_slabs_equal_the_one_shot_generator():
    from pyimsegm_amd.utilities.synthetic import config5_volume, ellipsoid_volume
    for shape in ((6, 40, 52), (9, 33, 64)):
        rng = np.random.default_rng(5)
        vol = ellipsoid_volume(shape).astype(np.float32)
        vol += (0.05 * rng.standard_normal(shape, dtype=np.float32))
        assert np.array_equal(vol, config5_volume(shape))


def test_config3_reference_run_shares_the_superpixels_of_config2():
    c2, c3 = bench.load_golden('reference_2048.npz'), bench.load_golden('reference_c3.npz')
    assert int(c2['image_crc']) == int(c3['image_crc']) and int(c2['slic_crc']) == int(c3['slic_crc'])
    assert c3['features'].shape == (int(c3['nb_superpixels']), 180) and len(c3['names']) == 180
    assert np.isfinite(c3['features']).all()


@pytest.mark.parametrize('seed', [100, 131, 163])
def test_config4_oracle_chain_equals_the_reference_run(oracle, seed):
    """imsegm/pipelines.py:160-241 under the group model of the reference's run over the 64 images"""
    from pyimsegm_amd.utilities.synthetic import voronoi_image
    ref = bench.load_golden('reference_c4.npz')
    i = list(ref['seeds']).index(seed)
    image = voronoi_image(*bench.C4_SHAPE, seed=seed)
    assert bench.crc32(image, np.uint8) == int(ref['image_crc'][i])
    _, segm, slic = bench._cpu_chain_color2d(image, bench.model_from_arrays(ref), bench.C4_SP_SIZE, bench.SP_REGUL, bench.GC_REGUL)
    assert bench.crc32(slic) == int(ref['slic_crc'][i]) and int(slic.max()) + 1 == int(ref['nb_superpixels'][i])
    assert bench.crc32(segm) == int(ref['segm_crc'][i])
    assert np.bincount(segm.ravel(), minlength=3).tolist() == ref['class_counts'][i].tolist()


def test_config5_oracle_chain_equals_the_reference_run(oracle):
    """imsegm/pipelines.py:382-431 on the 32 x 512 x 512 float32 volume: real scikit-image supervoxels + measure.label,
    the reference's gray statistics, and its graph cut under its own class probabilities"""
    from pyimsegm_amd import graph_cuts as gc
    from pyimsegm_amd.utilities.synthetic import config5_volume
    ref = bench.load_golden('reference_c5.npz')
    p = bench.C5_PARAMS
    vol = config5_volume(tuple(int(v) for v in ref['shape']))
    assert bench.crc32(vol, np.float32) == int(ref['volume_crc'])
    slic = oracle.segment_slic_img3d_gray(vol, p['sp_size'], p['sp_regul'], p['spacing'])
    assert bench.crc32(slic) == int(ref['slic_crc']) and int(slic.max()) + 1 == int(ref['nb_supervoxels'])
    seg32 = slic.astype(np.int32)
    mean = oracle.gray3d_stat(vol, seg32, 'mean')
    std = np.sqrt(oracle.gray3d_stat(vol, seg32, 'var', mean.astype(np.float32)))
    energy = oracle.gray3d_stat(vol, seg32, 'energy')
    features = np.nan_to_num(np.stack([mean, std, energy], axis=1))
    assert np.allclose(features, ref['features'], rtol=1e-5, atol=1e-5)
    proba = ref['proba']
    _, edges = oracle.adjacency(slic)
    edges = np.array(edges, dtype=np.int32)
    assert len(edges) == int(ref['nb_edges']) and bench.crc32(edges) == int(ref['edges_crc'])
    weights = gc.compute_edge_model(edges, proba, 'lT')
    weights = np.clip(weights / gc.compute_spatial_dist(oracle.centers(slic), edges, relative=True), 1e-3, 1e3)
    labels = oracle.cut_general_graph(edges, weights, gc.compute_unary_cost(proba), gc.compute_pairwise_cost(p['gc_regul'], proba.shape),
                                      n_iter=-1)
    assert np.array_equal(labels, ref['graph_labels'])
    assert bench.crc32(np.asarray(labels)[slic]) == int(ref['segm_crc'])
