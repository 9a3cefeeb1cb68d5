"""Compile the reference's own ``imsegm/features_cython.pyx`` (read in place from the reference
tree, never copied into the repo) into ``oracle/_ref/`` with the flags of the reference's
``setup.py:86-93``.  TEST INFRASTRUCTURE ONLY: used to pin the C restatement in
``imsegm_oracle.c`` and as the "reference" CPU descriptor baseline."""
import os
import shutil
import subprocess
import sys
import sysconfig
import tempfile

import numpy


def main(ref_root):
    here = os.path.dirname(os.path.abspath(__file__))
    out_dir = os.path.join(here, '_ref')
    os.makedirs(out_dir, exist_ok=True)
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
