"""SURVEY 8(f) rank 4 / rank 1 ON THE DEVICE: the reference's unchanged `run_eval_superpixels.py` (`--slico`),
`run_segm_slic_classif_graphcut.py` and `imsegm/ellipse_fitting.py` with the HIP kernels behind the overlay package
(tests/overlay_consumers_run.py --device under the image's conda Python 3.9; the GPU box has no /root/reference, the files come
from the bundle oracle/build_ref.py stages into oracle/_ref/reference).  One run, three tests; the log goes to gpurun_out/."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import test_overlay_consumers as cons  # noqa: E402

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(not os.path.exists(cons.PY39), reason="needs the image's conda interpreter (the reference's imports)")]


def _reference_tree():
    for cand in ('/root/reference', os.path.join(ROOT, 'oracle', '_ref', 'reference')):
        if all(os.path.isfile(os.path.join(cand, 'experiments_segmentation', f))
               for f in ('run_eval_superpixels.py', 'run_segm_slic_classif_graphcut.py')) \
                and os.path.isfile(os.path.join(cand, 'imsegm', 'ellipse_fitting.py')):
            return cand
    return None


@pytest.fixture(scope='module')
def seen(tmp_path_factory):
    ref = _reference_tree()
    assert ref is not None, 'no reference tree and no oracle/_ref/reference bundle: run __graft_entry__.build() where /root/reference exists'
    extra = {}
    # the conda interpreter ships a libstdc++ older than the one libamdhip64.so.7 needs ("conda interpreters")
    for cand in ('/usr/lib/x86_64-linux-gnu/libstdc++.so.6', '/usr/lib64/libstdc++.so.6'):
        if os.path.exists(cand):
            extra['LD_PRELOAD'] = (cand + ' ' + os.environ.get('LD_PRELOAD', '')).strip()
            break
    res, seen = cons.run_consumers(ref, tmp_path_factory.mktemp('consumers'), device=True, env_extra=extra)
    try:
        os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
        with open(os.path.join(ROOT, 'gpurun_out', 'overlay_consumers_device.log'), 'w') as fp:
            fp.write('$ /opt/conda/bin/python3.9 tests/overlay_consumers_run.py %s <tmp> --device\nexit code %d\n--- stdout\n%s\n--- stderr (tail)\n%s\n'
                     % (ref, res.returncode, res.stdout[-6000:], res.stderr[-6000:]))
    except OSError:
        pass
    assert res.returncode == 0 and seen is not None, res.stderr[-3000:]
    assert seen['library'] == os.path.join('pyimsegm_amd', 'libimsegm_hip.so')
    cons.check_consumers(seen)
    return seen


def test_run_eval_superpixels_slic_and_slico_on_the_device(seen):
    """/root/reference/experiments_segmentation/run_eval_superpixels.py:108-131: SLIC and SLICO reach the library, and the mean
    boundary distances are those of the oracle's label maps (bit-exact label maps -> equal distances)"""
    assert seen['device_calls']['slic'] >= 1 and seen['device_calls']['slico'] >= 1, seen['device_calls']
    assert abs(seen['eval_mean_boundary_distance']['slic'] - 1.468715196957587) < 1e-9
    assert abs(seen['eval_mean_boundary_distance']['slico'] - 2.049611423139335) < 1e-9


def test_run_segm_slic_classif_graphcut_on_the_device(seen):
    """/root/reference/experiments_segmentation/run_segm_slic_classif_graphcut.py:184-228 (SLIC, superpixel x annotation
    histogram, descriptors) and :323-385 (`segment_image` with the trained classifier: descriptors, graph, cut, gathers)"""
    calls = seen['device_calls']
    assert calls['slic'] >= 4 and calls['label_hist'] >= 2 and calls['segment_or_cut'] >= 1, calls
    assert seen['segment_classes'] == [0, 1, 2, 3]


def test_ellipse_fitting_slic_points_on_the_device(seen):
    """/root/reference/imsegm/ellipse_fitting.py:264-279, 625-645: SLIC of a gray segmentation, its centres (the fused graph /
    centres kernel), the boundary points of the module's doctest"""
    assert seen['ellipse_slic']['superpixels'] == 65 and seen['ellipse_slic']['centres'] == [65, 2]
    assert seen['ellipse_boundary_points'] == cons.EXPECTED_ELLIPSE_POINTS
