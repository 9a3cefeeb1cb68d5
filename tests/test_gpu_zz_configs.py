"""HIP path against the reference's own run AT THE SIZE of BASELINE configs 3 / 4 / 5 (tests/golden/reference_c{3,4,5}.npz,
produced by tests/golden/make_golden_configs.py with the unchanged reference, real scikit-image 0.18.3, scipy and
scikit-learn): the same checks `bench.py --config N` prints as `gpu_equals_reference_run`."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

pytestmark = pytest.mark.gpu


def test_config3_full_size_leung_malik_descriptors_and_segmentation():
    """/root/reference/imsegm/descriptors.py:1041-1106 on the 2048 x 2048 benchmark image: K x 180 descriptors within 1e-5"""
    from pyimsegm_amd import pipelines as pipe
    from pyimsegm_amd.utilities.synthetic import voronoi_image
    verdict = bench.compare_config3(voronoi_image(2048, 2048, seed=1), pipe)
    assert verdict is not None and verdict['gpu_equals_reference_run'], verdict


@pytest.mark.parametrize('seeds', [(100, 101, 117), (140, 163)])
def test_config4_images_equal_the_reference_run(seeds):
    """/root/reference/experiments_segmentation/run_segm_slic_model_graphcut.py:476-514 under the reference run's group model:
    the one-call pipeline and the staged one give the reference's superpixel map and segmentation, image by image"""
    from pyimsegm_amd import pipelines as pipe
    from pyimsegm_amd.descriptors import FEATURES_SET_COLOR
    from pyimsegm_amd.utilities.synthetic import voronoi_image
    ref = bench.load_golden('reference_c4.npz')
    model = bench.model_from_arrays(ref)
    for seed in seeds:
        i = list(ref['seeds']).index(seed)
        image = voronoi_image(*bench.C4_SHAPE, seed=seed)
        segm, _ = pipe._segment_color2d_one_call(image, model, FEATURES_SET_COLOR, bench.C4_SP_SIZE, bench.SP_REGUL, bench.GC_REGUL,
                                                 bench.EDGE_TYPE, want_soft=False, reuse=True)
        assert bench.crc32(segm) == int(ref['segm_crc'][i]), seed
        slic, features = pipe.compute_color2d_superpixels_features(image, FEATURES_SET_COLOR, bench.C4_SP_SIZE, bench.SP_REGUL)
        assert bench.crc32(slic) == int(ref['slic_crc'][i]), seed
        lo, hi = ref['features_offsets'][i], ref['features_offsets'][i + 1]
        assert np.allclose(features, ref['features'][lo:hi], rtol=1e-5, atol=1e-5)


def test_config5_reduced_volume_equals_the_reference_run():
    """/root/reference/imsegm/superpixels.py:93-111 + pipelines.py:382-431 on a float32 volume of 2 x 32 x 8 bricks"""
    from pyimsegm_amd import pipelines as pipe
    verdict = bench.compare_config5(bench.C5_REDUCED, pipe)
    assert verdict is not None and verdict['gpu_slic_equals_scikit_image'] and verdict['gpu_equals_reference_run'], verdict


@pytest.mark.skipif(not (os.path.isdir('/root/reference/imsegm') and os.path.exists('/opt/conda/bin/python3.9')),
                    reason='needs the reference tree and the conda interpreter of the build container')
def test_unchanged_reference_driver_on_the_device(tmp_path):
    """tests/overlay_driver_run.py --device: the reference's unchanged run_segm_slic_model_graphcut.py with the kernels
    (tests/test_overlay_driver.py is the same run with the oracle standing in for them)"""
    import json
    env = dict(os.environ, MPLBACKEND='Agg', OMP_NUM_THREADS='1')
    env.pop('PYTHONPATH', None)
    res = subprocess.run(['/opt/conda/bin/python3.9', os.path.join(ROOT, 'tests', 'overlay_driver_run.py'), '/root/reference', str(tmp_path),
                          '--device'], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, universal_newlines=True, timeout=900)
    assert res.returncode == 0, res.stderr[-3000:]
    seen = json.loads([ln for ln in res.stdout.splitlines() if ln.startswith('OVERLAY ')][-1][len('OVERLAY '):])
    assert seen['pipelines_is_hip'] and seen['shape'] == [900, 1200] and len(seen['classes']) > 1
    assert seen['region_growing_pixels'] is True
