"""The `imsegm` overlay package lets the reference's UNCHANGED driver `run_segm_slic_model_graphcut.py` import and run on
top of this repo (north_star "drops into run_segm_slic_model_graphcut.py unchanged").

Needs the reference tree and an interpreter that carries its dependencies (scikit-image 0.18, matplotlib, pandas): the
build container has both (/root/reference, conda Python 3.9); elsewhere the test is skipped.  Without a GPU the
kernels are stood in for by the oracle (tests/dryrun_plugin.py) -- what is tested here is the import graph and the glue:
`imsegm.pipelines` / `descriptors` / `labeling` are this repo's modules, `imsegm.utilities.{data_io,drawing,experiments}`
and `imsegm.region_growing` are the reference's own files, a name only the reference defines falls back to it."""
import json
import os
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
REF = '/root/reference'
PY39 = '/opt/conda/bin/python3.9'


@pytest.mark.skipif(not (os.path.isdir(os.path.join(REF, 'imsegm')) and os.path.exists(PY39)),
                    reason='needs the reference tree and the conda interpreter of the build container')
def test_unchanged_reference_driver_runs_on_the_overlay(tmp_path):
    env = dict(os.environ, MPLBACKEND='Agg', OMP_NUM_THREADS='1')
    env.pop('PYTHONPATH', None)
    env.pop('IMSEGM_REFERENCE', None)
    res = subprocess.run([PY39, os.path.join(HERE, 'overlay_driver_run.py'), REF, str(tmp_path)], env=env,
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, universal_newlines=True, timeout=900)
    assert res.returncode == 0, res.stderr[-3000:]
    line = [ln for ln in res.stdout.splitlines() if ln.startswith('OVERLAY ')][-1]
    seen = json.loads(line[len('OVERLAY '):])
    assert seen['pipelines_is_hip'] and seen['use_cython_written']
    assert seen['labeling'] == 'pyimsegm_amd.labeling'
    assert seen['data_io'] == 'imsegm/utilities/data_io.py' and seen['drawing'] == 'imsegm/utilities/drawing.py'
    assert seen['experiments'] == 'imsegm/utilities/experiments.py' and seen['region_growing'] == 'imsegm/region_growing.py'
    assert seen['fallback_attr'] == 'imsegm._reference.descriptors' and seen['gco'] == 'gco/__init__.py'
    # the reference's own region_growing module with its graph cuts going through the gco shim: gco's known answers
    assert seen['region_growing_pixels'] is True
    assert seen['region_growing_slic'] == [[0, 0, 0, 0, 0, 1, 1, 1, 1, 0]] * 2
    assert seen['shape'] == [900, 1200] and len(seen['classes']) > 1          # a real segmentation, not the except branch
    assert seen['files'] == ['0000_img_12.npz', '0000_img_12.png']
    assert seen['visu'] == ['0000_img_12.png', '0000_img_12_debug.png']       # incl. figure_segm_graphcut_debug


def test_overlay_is_inert_without_a_reference():
    """stand-alone (no reference installed): the alias package exposes this repo's modules only"""
    code = ("import sys; sys.path.insert(0, %r); import imsegm, imsegm.pipelines, imsegm.utilities.data_io as io; "
            "import pyimsegm_amd.utilities.data_io as own; assert imsegm.REFERENCE_PATH is None; assert io is own; "
            "import importlib\n"
            "try:\n    importlib.import_module('imsegm.region_growing'); raise SystemExit(3)\n"
            "except ImportError:\n    pass\n"
            "try:\n    imsegm.descriptors.reconstruct_ray_features_2d; raise SystemExit(4)\n"
            "except AttributeError:\n    pass\n") % os.path.dirname(HERE)
    import sys
    res = subprocess.run([sys.executable, '-c', code], cwd='/', stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                         universal_newlines=True, timeout=300)
    assert res.returncode == 0, res.stderr[-2000:]
