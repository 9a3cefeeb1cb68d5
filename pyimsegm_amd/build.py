"""Build libimsegm_hip.so (hand-written HIP for gfx950) in-tree with hipcc.

    python -m pyimsegm_amd.build            # incremental build
    python -m pyimsegm_amd.build --force

The shared library is git-ignored (history stays source-only) but travels with the working tree.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIB = os.path.join(HERE, 'libimsegm_hip.so')
SOURCES = ['api.hip', 'api_image2d.hip', 'api_texture.hip', 'api_volume.hip', 'api_fused.hip', 'api_natives.hip', 'batch.hip', 'slic_pre.hip', 'slic.hip', 'connectivity.hip', 'stats.hip', 'graph.hip', 'graphcut.hip', 'texture.hip', 'volume.hip', 'terms.hip', 'natives.hip', 'median.hip', 'output.hip']
HEADERS = ['common.h', 'slic.h', 'session.h', os.path.join('..', '..', 'include', 'imsegm_hip.h')]
# -ffp-contract=off: every fp64 operation rounds on its own -- the bit-exactness contract with the
# CPU oracle; no fast-math anywhere.
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-ffp-contract=off', '-fvisibility=hidden',
         '-Wall', '-Wno-unused-function', '-Wno-pass-failed']


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    objdir = os.path.join(HERE, 'build')
    os.makedirs(objdir, exist_ok=True)
    headers = [os.path.join(CSRC, h) for h in HEADERS]
    objs = []
    procs = []
    for src in SOURCES:
        spath = os.path.join(CSRC, src)
        obj = os.path.join(objdir, src.replace('.hip', '.o'))
        objs.append(obj)
        if force or _stale(obj, [spath] + headers):
            cmd = [hipcc] + FLAGS + ['-c', spath, '-o', obj]
            if verbose:
                print(' '.join(cmd), flush=True)
            procs.append((src, subprocess.Popen(cmd)))
    for src, p in procs:
        if p.wait() != 0:
            raise RuntimeError('hipcc failed on %s' % src)
    if force or procs or _stale(LIB, objs):
        cmd = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + objs
        if verbose:
            print(' '.join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == '__main__':
    build(force='--force' in sys.argv)
