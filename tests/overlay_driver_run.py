#!/usr/bin/env python
"""Run the reference's UNCHANGED experiment driver (`experiments_segmentation/run_segm_slic_model_graphcut.py`) on top of
this repo's `imsegm` overlay package (test infrastructure; started by tests/test_overlay_driver.py under an interpreter
that carries the reference's dependencies -- the build container's conda Python 3.9):

    /opt/conda/bin/python3.9 tests/overlay_driver_run.py /root/reference <out_dir> [--device]

* `sys.path` = [this repo, the reference tree]: `import imsegm` is this repo's overlay, which finds the reference package
  behind it and completes itself with the reference's own `utilities.{data_io,drawing,experiments}`, `labeling`
  fall-backs etc.; the driver module is imported by its file name, nothing of it is edited or copied.
* `gco` is this repo's shim (gco/__init__.py); nibabel / planar / OleFileIO_PL (file readers, ellipse drawing) are absent from the
  container: empty stub modules, exactly as tests/golden/make_golden_reference.py does.  They are never called here.
  The `np.float` / `np.int` / `np.bool` aliases the reference's drawing module still uses are restored (numpy >= 1.24).
* without `--device` (no GPU in the build container) the ctypes session classes are replaced by the oracle-backed
  stand-ins of tests/dryrun_plugin.py -- this run then checks the IMPORT GRAPH and the Python glue, not the kernels.
Prints one JSON line with what it saw.
"""
import json
import os
import sys
import types
import warnings

warnings.filterwarnings('ignore')
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def main():
    ref, out_dir = os.path.abspath(sys.argv[1]), os.path.abspath(sys.argv[2])
    use_device = '--device' in sys.argv[3:]
    sys.path[:0] = [ROOT, ref, os.path.join(ref, 'experiments_segmentation')]
    for name in ('nibabel', 'planar', 'OleFileIO_PL'):
        if name not in sys.modules:
            try:
                __import__(name)
            except ImportError:
                sys.modules[name] = types.ModuleType(name)
    if not hasattr(sys.modules['planar'], 'line'):
        sys.modules['planar'].line = types.ModuleType('planar.line')
    os.chdir(ref)                                   # the driver resolves 'data-images' relative to the tree
    if not use_device:
        sys.path.insert(0, HERE)
        import dryrun_plugin
        dryrun_plugin.pytest_configure(None)
    import numpy as np
    for alias, typ in (('float', float), ('int', int), ('bool', bool)):     # the reference's drawing module predates
        if alias not in np.__dict__:                                        # numpy 1.24 (np.float was removed there)
            setattr(np, alias, typ)
    import imsegm
    assert imsegm.REFERENCE_PATH == os.path.join(ref, 'imsegm'), imsegm.REFERENCE_PATH
    device_calls = [0]
    if use_device:                                  # count the SLIC runs that reach libimsegm_hip.so
        from pyimsegm_amd import _hip
        real_slic = _hip.Image2D.slic

        def counted_slic(self, *a, **kw):
            device_calls[0] += 1
            return real_slic(self, *a, **kw)
        _hip.Image2D.slic = counted_slic
    import run_segm_slic_model_graphcut as drv     # the reference's driver, unchanged
    import pyimsegm_amd.pipelines

    seen = {
        'pipelines_is_hip': drv.seg_pipe is pyimsegm_amd.pipelines,
        'use_cython_written': drv.seg_fts.USE_CYTHON is False,
        'data_io': os.path.relpath(drv.tl_data.__file__, ref),
        'drawing': os.path.relpath(drv.tl_visu.__file__, ref),
        'experiments': os.path.relpath(drv.tl_expt.__file__, ref),
        'labeling': drv.seg_lbs.__name__,
    }
    # a name of a shadowed module that only the reference defines resolves to the reference's function
    import imsegm.descriptors as seg_fts
    seen['fallback_attr'] = seg_fts.reconstruct_ray_features_2d.__module__
    import gco                                      # this repo's shim: the reference's region_growing cuts graphs on the device
    seen['gco'] = os.path.relpath(gco.__file__, ROOT)
    import imsegm.region_growing                    # noqa: F401  (reference module importing shadowed ones by name)
    seen['region_growing'] = os.path.relpath(sys.modules['imsegm.region_growing'].__file__, ref)

    # the reference's OWN region_growing functions cutting their graphs through the shim (doctests region_growing.py:72-75,
    # 187-200: the known answers of gco there)
    import imsegm.region_growing as seg_rg
    sys.path.insert(0, HERE)
    import natives_cases as NC
    got = seg_rg.object_segmentation_graphcut_pixels(NC.GRID_SEGM, NC.GRID_CENTRES, gc_regul=0., coef_shape=0.5)
    got2 = seg_rg.object_segmentation_graphcut_pixels(NC.GRID_SEGM, NC.GRID_CENTRES, gc_regul=.5, seed_size=1)
    seen['region_growing_pixels'] = bool(np.array_equal(got, NC.GRID_EXPECT_SHAPE) and np.array_equal(got2, NC.GRID_EXPECT_SEED))
    slic = np.array([[0] * 3 + [1] * 3 + [2] * 3 + [3] * 3 + [4] * 3, [5] * 3 + [6] * 3 + [7] * 3 + [8] * 3 + [9] * 3])
    segm = np.array([[0] * 15, [1] * 12 + [0] * 3])
    seen['region_growing_slic'] = [
        np.asarray(seg_rg.object_segmentation_graphcut_slic(slic, segm, [(1, 7)], gc_regul=0., edge_coef=1., coef_shape=1.)).tolist(),
        np.asarray(seg_rg.object_segmentation_graphcut_slic(slic, segm, [(1, 7)], gc_regul=1., edge_coef=1.)).tolist()]

    params = dict(drv.SEGM_PARAMS)
    params['path_exp'] = out_dir
    for sub in (drv.FOLDER_IMAGE, drv.FOLDER_SEGM_GMM, drv.FOLDER_SEGM_GMM_VISU):
        os.makedirs(os.path.join(out_dir, sub), exist_ok=True)
    path_img = os.path.join(ref, 'data-images', 'drosophila_disc', 'image', 'img_12.jpg')
    np.random.seed(0)
    name, segm = drv.segment_image_independent((0, path_img), params, os.path.join(out_dir, drv.FOLDER_SEGM_GMM),
                                               os.path.join(out_dir, drv.FOLDER_SEGM_GMM_VISU), show_debug_imgs=True)
    seen.update(name=name, shape=list(segm.shape), classes=sorted(int(v) for v in np.unique(segm)),
                files=sorted(os.listdir(os.path.join(out_dir, drv.FOLDER_SEGM_GMM))),
                visu=sorted(os.listdir(os.path.join(out_dir, drv.FOLDER_SEGM_GMM_VISU))))
    seen['device_calls'] = device_calls[0]
    if use_device:
        from pyimsegm_amd import _hip
        seen['library'] = os.path.relpath(_hip.LIB_PATH, ROOT)
    print('OVERLAY ' + json.dumps(seen))


if __name__ == '__main__':
    main()
