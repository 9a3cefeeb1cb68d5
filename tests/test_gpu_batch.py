"""Several images of one size per launch chain (csrc/batch.hip, imsegm_batch2d_*: image = blockIdx.z in every kernel) against
the same images one at a time (imsegm_image2d_run_color), the CPU oracle and the reference's own run on the 64 images of BASELINE
config 4 -- what the reference does by mapping `segment_image_model` over a process pool
(/root/reference/experiments_segmentation/run_segm_slic_model_graphcut.py:451-473, 505-514).  Bit-exact: label maps and
segmentations are integer results."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def hip():
    from pyimsegm_amd import _hip
    return _hip


def _model(images, sp_size, sp_regul, nb_classes=3):
    from pyimsegm_amd import pipelines as pipe
    from pyimsegm_amd.descriptors import FEATURES_SET_COLOR
    from pyimsegm_amd.graph_cuts import estim_class_model
    np.random.seed(0)
    res = pipe._ResidentImage(images[0], FEATURES_SET_COLOR, sp_size, sp_regul)
    model = estim_class_model(res.features, nb_classes, 'GMM', None, True)
    res.close()
    return model


def _one_by_one(images, model, sp_size, sp_regul, gc_regul=2.0, edge='model'):
    from pyimsegm_amd import pipelines as pipe
    from pyimsegm_amd.descriptors import FEATURES_SET_COLOR
    out = []
    for im in images:
        got = pipe._segment_color2d_one_call(im, model, FEATURES_SET_COLOR, sp_size, sp_regul, gc_regul, edge, want_soft=False, reuse=True)
        assert got is not None
        out.append(np.array(got[0]))
    return out


def _batched(images, model, sp_size, sp_regul, gc_regul=2.0, edge='model', keep=None):
    from pyimsegm_amd import pipelines as pipe
    from pyimsegm_amd.descriptors import FEATURES_SET_COLOR
    got = pipe._segment_color2d_batch_call(images, model, FEATURES_SET_COLOR, sp_size, sp_regul, gc_regul, edge, with_batch=keep)
    assert got is not None
    return [np.array(a) for a in got]


@pytest.mark.parametrize('shape,sp_size,count', [((200, 300), 18, 3), ((97, 130), 12, 5), ((330, 257), 25, 2), ((64, 64), 10, 8)])
def test_batch_equals_one_image_at_a_time(hip, oracle, shape, sp_size, count):
    """ragged sizes (partial tiles in both directions), fewer images than the batch holds, superpixel maps and segmentations"""
    from pyimsegm_amd.utilities.synthetic import voronoi_image
    images = [voronoi_image(shape[0], shape[1], seed=40 + i) for i in range(count)]
    model = _model(images, sp_size, 0.2)
    single = _one_by_one(images, model, sp_size, 0.2)
    maps = []
    both = _batched(images, model, sp_size, 0.2, keep=lambda b: maps.extend(b.get_labels(i) for i in range(count)))
    for i in range(count):
        assert np.array_equal(both[i], single[i]), i
        assert np.array_equal(maps[i], oracle.segment_slic_img2d(images[i], sp_size, 0.2)), i
    assert len(np.unique(both[0])) > 1


def test_batch_object_is_recycled_and_takes_new_parameters(hip):
    """the arena of a batch is kept by the thread's context; a second batch of the same size reuses it, other SLIC parameters
    (another layout of the slices) make it start from a zeroed arena again; results stay those of the single-image path"""
    from pyimsegm_amd.utilities.synthetic import voronoi_image
    images = [voronoi_image(150, 210, seed=7 + i) for i in range(4)]
    model = _model(images, 15, 0.2)
    a = _batched(images, model, 15, 0.2)
    b = _batched(images[::-1], model, 15, 0.2)
    assert all(np.array_equal(x, y) for x, y in zip(a, b[::-1]))
    c = _batched(images[:2], model, 22, 0.35, gc_regul=0.5, edge='spatial')
    want = _one_by_one(images[:2], model, 22, 0.35, gc_regul=0.5, edge='spatial')
    assert all(np.array_equal(x, y) for x, y in zip(c, want))
    d = _batched(images, model, 15, 0.2)
    assert all(np.array_equal(x, y) for x, y in zip(a, d))
    # gc_regul = 0: the argmin of the unary cost instead of the cut (graph_cuts.py:729-731)
    e = _batched(images[:3], model, 15, 0.2, gc_regul=0.)
    want = _one_by_one(images[:3], model, 15, 0.2, gc_regul=0.)
    assert all(np.array_equal(x, y) for x, y in zip(e, want))


def test_batch_of_float64_images(hip):
    from pyimsegm_amd.utilities.synthetic import voronoi_image
    images = [voronoi_image(120, 160, seed=3 + i).astype(np.float64) / 255. for i in range(3)]
    model = _model(images, 14, 0.25)
    both = _batched(images, model, 14, 0.25)
    single = _one_by_one(images, model, 14, 0.25)
    assert all(np.array_equal(x, y) for x, y in zip(both, single))


def test_batch_with_a_map_that_leaves_the_tile_path(hip):
    """one image of the batch is noise under a weak regularisation: its k-means assignment falls into more local components per
    64 x 32 tile than the tile path of the connectivity stage has slots for, so THAT image goes through the general path alone
    (its own slice of the arena) while the others stay on the batched tile path"""
    from pyimsegm_amd.utilities.synthetic import voronoi_image
    rng = np.random.default_rng(12)
    noisy = rng.integers(0, 256, (192, 256, 3), dtype=np.uint8)
    images = [voronoi_image(192, 256, seed=21), noisy, voronoi_image(192, 256, seed=22)]
    model = _model(images, 20, 0.2)
    before = hip.load_library().imsegm_debug_conn_general_runs()
    both = _batched(images, model, 20, 0.02)
    during = hip.load_library().imsegm_debug_conn_general_runs()
    single = _one_by_one(images, model, 20, 0.02)
    assert all(np.array_equal(x, y) for x, y in zip(both, single))
    assert during > before, 'the noise image was expected to leave the tile path (else this test does not test the hand-over)'


def test_config4_batches_equal_the_reference_run(hip):
    """BASELINE configs[3]: the images of 647 x 1024 eight at a time under the group model of the reference's own run
    (tests/golden/reference_c4.npz): superpixel map and segmentation of every image, by CRC"""
    from pyimsegm_amd.utilities.synthetic import voronoi_image
    ref = bench.load_golden('reference_c4.npz')
    model = bench.model_from_arrays(ref)
    seeds = [100, 101, 117, 131, 140, 150, 162, 163]
    images = [voronoi_image(*bench.C4_SHAPE, seed=s) for s in seeds]
    maps = []
    got = _batched(images, model, bench.C4_SP_SIZE, bench.SP_REGUL, bench.GC_REGUL, bench.EDGE_TYPE,
                   keep=lambda b: maps.extend(b.get_labels(i) for i in range(len(seeds))))
    for i, seed in enumerate(seeds):
        j = list(ref['seeds']).index(seed)
        assert bench.crc32(maps[i]) == int(ref['slic_crc'][j]), seed
        assert bench.crc32(got[i]) == int(ref['segm_crc'][j]), seed


def test_batch_refuses_what_it_does_not_take(hip):
    from pyimsegm_amd import pipelines as pipe
    from pyimsegm_amd.descriptors import FEATURES_SET_COLOR
    from pyimsegm_amd.utilities.synthetic import voronoi_image
    images = [voronoi_image(80, 90, seed=1), voronoi_image(80, 91, seed=2)]
    model = _model(images[:1], 12, 0.2)
    # images of different sizes, features the device does not evaluate: the caller takes the images one by one
    assert pipe._segment_color2d_batch_call(images, model, FEATURES_SET_COLOR, 12, 0.2, 2.0, 'model') is None
    assert pipe._segment_color2d_batch_call(images[:1], model, {'color': ('mean', 'median')}, 12, 0.2, 2.0, 'model') is None
    batch = hip.Batch2D(2, 80, 90)
    with pytest.raises(ValueError):
        batch.run_color([images[0]] * 3, 40, 10., pipe._device_gmm(model), np.ones((3, 3)) - np.eye(3))
    with pytest.raises(hip.HipError):      # sigma 3 -> radius 12: the three-pass pre-processing, single images only
        batch.run_color([images[0]], 40, 10., pipe._device_gmm(model), np.ones((3, 3)) - np.eye(3), sigma=3.)
    batch.close()
