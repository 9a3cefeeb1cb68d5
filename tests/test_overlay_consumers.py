"""The other consumers of the hot path among the reference's unchanged files -- `run_eval_superpixels.py` (SLIC / SLICO),
`run_segm_slic_classif_graphcut.py` (supervised path) and `imsegm/ellipse_fitting.py` -- import and run on the `imsegm` overlay
package.  Here, without a GPU, the oracle stands in for the kernels (tests/dryrun_plugin.py): what is
tested is the import graph and the glue; tests/test_gpu_zz_consumers.py is the same run on the device."""
import json
import os
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
REF = '/root/reference'
PY39 = '/opt/conda/bin/python3.9'

#: what the run must report whatever stands behind the ctypes layer (the HIP kernels equal the oracle bit for bit on label maps)
EXPECTED_ELLIPSE_POINTS = {'count': 14, 'first': [6, 85], 'second': [8, 150], 'last': [92, 118]}     # ellipse_fitting.py:631-637


def check_consumers(seen):
    assert seen['eval_superpixels_is_hip'] and seen['classif_driver_is_hip'] and seen['ellipse_fitting_is_hip']
    # 1. run_eval_superpixels.py:108-131 -- a finite mean boundary distance for SLIC and for SLICO, and not the same one
    dist = seen['eval_mean_boundary_distance']
    assert seen['eval_name'] == 'insitu7545' and 0.5 < dist['slic'] < 5. and 0.5 < dist['slico'] < 5. and dist['slic'] != dist['slico']
    # 2. run_segm_slic_classif_graphcut.py:184-228, 323-385
    assert seen['train_images'] == {'0000_insitu4174': [659, 1033], '0001_insitu7545': [647, 1024]}
    assert seen['feature_names'] == ['gray_mean', 'gray_std', 'gray_energy']
    assert seen['segment_shape'] == [647, 1024] and len(seen['segment_classes']) >= 3
    assert seen['segment_files'] == ['0000_insitu7545.npz', '0000_insitu7545.png', '0000_insitu7545_MAP.png']
    assert seen['segment_agrees_with_annot'] > 0.8 and seen['map_agrees_with_annot'] > 0.8      # a trained image: mostly right
    # 3. ellipse_fitting.py:264-279, 625-645 -- the points the module's own doctest prints
    assert seen['ellipse_fitting'] == 'imsegm/ellipse_fitting.py'
    assert seen['ellipse_slic']['shape'] == [100, 200] and seen['ellipse_slic']['labels_found'] == [0, 1]
    assert seen['ellipse_boundary_points'] == EXPECTED_ELLIPSE_POINTS


def run_consumers(ref, out_dir, device, env_extra=None):
    env = dict(os.environ, MPLBACKEND='Agg', OMP_NUM_THREADS='1')
    env.pop('PYTHONPATH', None)
    env.pop('IMSEGM_REFERENCE', None)
    env.update(env_extra or {})
    cmd = [PY39, os.path.join(HERE, 'overlay_consumers_run.py'), ref, str(out_dir)] + (['--device'] if device else [])
    res = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, universal_newlines=True, timeout=1200)
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith('CONSUMERS ')]
    return res, (json.loads(lines[-1][len('CONSUMERS '):]) if lines else None)


@pytest.mark.skipif(not (os.path.isdir(os.path.join(REF, 'imsegm')) and os.path.exists(PY39)),
                    reason='needs the reference tree and the conda interpreter of the build container')
def test_other_consumers_of_the_reference_run_on_the_overlay(tmp_path):
    res, seen = run_consumers(REF, tmp_path, device=False)
    assert res.returncode == 0 and seen is not None, res.stderr[-3000:]
    check_consumers(seen)
    # the figures the device run must reproduce (tests/test_gpu_zz_consumers.py): the oracle's label maps are the kernels'
    assert abs(seen['eval_mean_boundary_distance']['slic'] - 1.468715196957587) < 1e-9
    assert abs(seen['eval_mean_boundary_distance']['slico'] - 2.049611423139335) < 1e-9
    assert seen['ellipse_slic']['superpixels'] == 65
