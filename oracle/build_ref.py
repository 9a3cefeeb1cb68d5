"""Compile the reference's own ``imsegm/features_cython.pyx`` (read in place from the reference
tree, never copied into the repo) into ``oracle/_ref/`` with the flags of the reference's
``setup.py:86-93``, and stage the files the reference's unchanged experiment driver needs
(``stage_driver_bundle``) next to it.  TEST INFRASTRUCTURE ONLY: used to pin the C restatement in
``imsegm_oracle.c``, as the "reference" CPU descriptor baseline, and to run the reference's driver on
the GPU box, where /root/reference does not exist."""
import os
import shutil
import subprocess
import sys
import sysconfig
import tempfile

import numpy


#: what the reference's unchanged experiment driver needs to run on top of this repo's overlay package on a box WITHOUT the
#: reference tree (the GPU box): the driver script, the reference package behind the overlay (utilities, region growing, ...: the
#: modules the overlay does not shadow) and the one sample image the test segments
DRIVER_BUNDLE = (
    ('experiments_segmentation', ('run_segm_slic_model_graphcut.py', 'run_segm_slic_classif_graphcut.py', 'run_eval_superpixels.py')),
    ('imsegm', None),                       # the whole package directory (*.py, *.pyx)
    (os.path.join('data-images', 'drosophila_disc', 'image'), ('img_12.jpg', )),
    # tests/overlay_consumers_run.py: two ovary slices with their annotations (the reference's own sample data)
    (os.path.join('data-images', 'drosophila_ovary_slice', 'image'), ('insitu4174.jpg', 'insitu7545.jpg')),
    (os.path.join('data-images', 'drosophila_ovary_slice', 'annot_struct'), ('insitu4174.png', 'insitu7545.png')),
    (os.path.join('data-images', 'drosophila_ovary_slice', 'annot_eggs'), ('insitu7545.png', )),
)


def stage_driver_bundle(ref_root, out_dir):
    """oracle/_ref/reference/: a byte-for-byte copy of the files named in DRIVER_BUNDLE, laid out as in the reference tree, so
    that `tests/overlay_driver_run.py <that directory> <out> --device` runs the UNCHANGED driver on the GPU box
    (tests/test_gpu_zz_configs.py::test_unchanged_reference_driver_on_the_device).  Like the compiled .pyx next to it this is a
    build output: git-ignored, never part of the repo's history, travels with the working tree."""
    dst_root = os.path.join(out_dir, 'reference')
    copied = 0
    for sub, names in DRIVER_BUNDLE:
        src_dir, dst_dir = os.path.join(ref_root, sub), os.path.join(dst_root, sub)
        if names is None:
            names = []
            for base, dirs, files in os.walk(src_dir):
                dirs[:] = [d for d in dirs if d != '__pycache__']
                names += [os.path.relpath(os.path.join(base, f), src_dir) for f in files if f.endswith(('.py', '.pyx'))]
        for name in names:
            src, dst = os.path.join(src_dir, name), os.path.join(dst_dir, name)
            os.makedirs(os.path.dirname(dst), exist_ok=True)
            if not os.path.exists(dst) or os.path.getmtime(dst) < os.path.getmtime(src) or os.path.getsize(dst) != os.path.getsize(src):
                shutil.copy2(src, dst)
                copied += 1
    print('oracle/_ref/reference: %d file(s) staged' % copied if copied else 'oracle/_ref/reference up to date')


def main(ref_root):
    here = os.path.dirname(os.path.abspath(__file__))
    out_dir = os.path.join(here, '_ref')
    os.makedirs(out_dir, exist_ok=True)
    stage_driver_bundle(ref_root, out_dir)
    pyx = os.path.join(ref_root, 'imsegm', 'features_cython.pyx')
    ext = sysconfig.get_config_var('EXT_SUFFIX')
    target = os.path.join(out_dir, 'features_cython' + ext)
    if os.path.exists(target) and os.path.getmtime(target) >= os.path.getmtime(pyx):
        print('oracle/_ref up to date')
        return
    with tempfile.TemporaryDirectory() as tmp:
        cpp = os.path.join(tmp, 'features_cython.cpp')
        subprocess.check_call([sys.executable, '-m', 'cython', '--cplus', '-3', pyx, '-o', cpp])
        cmd = [
            'g++', '-shared', '-fPIC', '-O3', '-ffast-math', '-march=x86-64-v2', '-w',
            '-DNPY_NO_DEPRECATED_API=NPY_1_7_API_VERSION',
            '-I' + numpy.get_include(), '-I' + sysconfig.get_paths()['include'],
            cpp, '-o', target,
        ]
        subprocess.check_call(cmd)
    print('built', target)


if __name__ == '__main__':
    main(sys.argv[1] if len(sys.argv) > 1 else '/root/reference')
