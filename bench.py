#!/usr/bin/env python
"""bench.py -- headline benchmark of the SLIC -> descriptors -> GraphCut hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--inflight M]

Workload (BASELINE.json configs[1]): one synthetic 2048 x 2048 RGB uint8 image per GPU, SLIC
(sp_size 46 -> n_segments 1982, K = 2025 grid centroids) + colour mean/std/energy descriptors +
3-class alpha-expansion GraphCut (gc_regul 2.0, edge type 'model') with a class model fitted once
during warm-up (the reference's `segment_color2d_slic_features_model_graphcut`,
imsegm/pipelines.py:160).  One step = one pass of that function over the resident image: the
input image is already in HBM when the timed region starts and the outputs (segm H x W int32,
segm_soft H x W x 3 float64) stay in HBM; the scikit-learn `predict_proba` and the numpy edge-weight
formulas of the reference run on the host inside the timed region, as do the K x F / E / K x C
transfers between the stages.  The K timed steps are taken by M worker threads per process (default: by K, 6 from 40 steps), each
with its own HIP stream and resident copy of the image, so that the host stages of one step overlap the
kernels of another -- the reference maps a pool of worker processes over the images.  The `roofline` and
`stage_ms_per_step` figures come from a second, un-overlapped pass on one stream (the assignment kernel is
timed by a HIP event pair attached to its dispatch).  N > 1: one process per GPU (torch.distributed / RCCL),
every rank segments its own image (weak scaling, no data-path collective) and every label map is gathered
on rank 0 with one RCCL gather per step, zero copy from HBM.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SP_SIZE, SP_REGUL, NB_CLASSES, GC_REGUL, EDGE_TYPE = 46, 0.2, 3, 2.0, 'model'
HEIGHT = WIDTH = 2048
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s
ASSIGN_BYTES_PER_PX = 28.0     # read fp64 Lab 3 x 8 B + write int32 label 4 B per pixel per sweep


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=100)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--size', type=int, default=HEIGHT, help='image edge (default: the BASELINE 2048)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--switch-interval', type=float, default=None, help='sys.setswitchinterval for the worker threads')
    ap.add_argument('--host-pool', type=int, default=1,
                    help='1: the numpy stages (class model, graph-cut terms) of the images in flight run in helper '
                         'processes (pyimsegm_amd.hostpool), 0: in the worker threads themselves')
    ap.add_argument('--inflight', type=int, default=0,
                    help='images in flight per GPU (worker threads, one HIP stream each; the reference runs a '
                         'pool of nb_workers processes over the images); 0 = by the number of timed steps: a deep '
                         'pipeline only pays off when its fill and drain are amortised (6 from 40 steps, else 4 or fewer)')
    return ap.parse_args()


def cpu_baseline(image, model, passes=5):
    """the CPU oracle (port of the reference path) timed on one host core: `passes` full images (about 12 s),
    median pass reported (warm-up free C code, median of 5)"""
    runs = [_cpu_baseline_pass(image, model) for _ in range(max(1, passes))]
    runs.sort(key=lambda r: r[0])
    total, parts, segm, soft = runs[len(runs) // 2]
    return {
        'value': round(image.shape[0] * image.shape[1] / total / 1e6, 4),
        'unit': 'Mpixels/s',
        'cores': 1,
        'kind': 'port',
        'sample': '%d passes over one full %dx%d image (%.1f s of CPU), median pass: oracle C SLIC %.2fs + descriptors '
                  '%.2fs + graph/weights %.2fs + GC %.3fs + gathers %.2fs'
                  % ((len(runs), image.shape[0], image.shape[1], sum(r[0] for r in runs)) + parts),
    }, segm, soft


_SKIMAGE_SCRIPT = r'''
import sys, time, warnings
warnings.filterwarnings("ignore")
import numpy as np
from skimage.segmentation import slic
import skimage
img = np.load(sys.argv[1])
sp_size, rc = float(sys.argv[3]), float(sys.argv[4])
nb_pixels = img.shape[0] * img.shape[1]
if img.min() != 0. or img.max() != 1.:                       # imsegm/superpixels.py:53-54
    img = (img - img.min()) / float(img.max() - img.min())
t0 = time.perf_counter()
labels = slic(img, n_segments=int(nb_pixels / sp_size**2), compactness=(sp_size * rc)**1.5, sigma=1,
              enforce_connectivity=True, slic_zero=False)      # imsegm/superpixels.py:57-63
t1 = time.perf_counter()
np.save(sys.argv[2], np.asarray(labels).astype(np.int32))
print("%s %.4f" % (skimage.__version__, t1 - t0))
'''


def real_skimage_slic(image):
    """the REAL `skimage.segmentation.slic` behind imsegm/superpixels.py:61-63, when the box carries the image's conda
    Python 3.9 with scikit-image (the interpreter of this script cannot import it): (version, seconds of the slic
    call on one host core, label map), or None.  Reported next to the oracle's own time; never part of `value`."""
    import subprocess
    import tempfile
    py = os.environ.get('IMSEGM_SKIMAGE_PYTHON', '/opt/conda/bin/python3.9')
    if not os.path.exists(py):
        return None
    try:
        with tempfile.TemporaryDirectory() as tmp:
            src, dst = os.path.join(tmp, 'image.npy'), os.path.join(tmp, 'labels.npy')
            np.save(src, image)
            run = subprocess.run([py, '-c', _SKIMAGE_SCRIPT, src, dst, str(SP_SIZE), str(SP_REGUL)], capture_output=True,
                                 text=True, timeout=300)
            if run.returncode != 0:
                return None
            version, seconds = run.stdout.split()[-2:]
            return version, float(seconds), np.load(dst)
    except Exception:
        return None


def compare_with_reference_run(image, sess, mode, pipe, predict_proba):
    """one extra, untimed GPU pass with the class model of the reference's own run on the benchmark image
    (tests/golden/reference_2048.npz: real scikit-image 0.18.3 + the reference's Cython descriptors + its scikit-learn
    GMM, gco bridged to the oracle; label maps stored as CRC32): the superpixel map and the final segmentation of the
    GPU must have the same checksums"""
    import zlib
    from sklearn.mixture import GaussianMixture
    from sklearn.pipeline import Pipeline
    from sklearn.preprocessing import StandardScaler
    path = os.path.join(ROOT, 'tests', 'golden', 'reference_2048.npz')
    if not os.path.exists(path):
        return None
    ref = np.load(path, allow_pickle=False)
    if zlib.crc32(np.ascontiguousarray(image).tobytes()) != int(ref['image_crc']):
        return None
    scaler = StandardScaler()
    scaler.mean_, scaler.scale_ = ref['scaler_mean'], ref['scaler_scale']
    scaler.var_, scaler.n_features_in_ = scaler.scale_**2, len(scaler.mean_)
    gmm = GaussianMixture(n_components=len(ref['gmm_weights']), covariance_type='full')
    gmm.weights_, gmm.means_ = ref['gmm_weights'], ref['gmm_means']
    gmm.covariances_, gmm.precisions_cholesky_ = ref['gmm_covariances'], ref['gmm_precisions_cholesky']
    gmm.precisions_ = np.array([pc @ pc.T for pc in gmm.precisions_cholesky_])
    gmm.converged_, gmm.n_features_in_ = True, scaler.n_features_in_
    model = Pipeline([('scaler', scaler), ('GMM', gmm)])
    from pyimsegm_amd.descriptors import FEATURES_SET_COLOR
    res = pipe._ResidentImage(image, FEATURES_SET_COLOR, SP_SIZE, SP_REGUL, session=(sess, mode))
    segm, _ = res.segment(predict_proba(model, res.features), GC_REGUL, EDGE_TYPE, to_host=True)
    slic_ok = zlib.crc32(np.ascontiguousarray(res.slic, dtype=np.int32).tobytes()) == int(ref['slic_crc'])
    segm_ok = zlib.crc32(np.ascontiguousarray(segm, dtype=np.int32).tobytes()) == int(ref['segm_crc'])
    return {'gpu_equals_reference_run': bool(slic_ok and segm_ok),
            'reference_run': 'tests/golden/reference_2048.npz: the reference itself on this image (%s), superpixel map %s, '
                             'segmentation %s' % (str(ref['versions']), 'equal' if slic_ok else 'DIFFERENT',
                                                  'equal' if segm_ok else 'DIFFERENT')}


def _cpu_baseline_pass(image, model):
    from oracle import oracle as orc
    from pyimsegm_amd import graph_cuts as gc
    orc.lib()
    t0 = time.perf_counter()
    slic = orc.segment_slic_img2d(image, SP_SIZE, SP_REGUL)
    t1 = time.perf_counter()
    img32 = np.asarray(image, dtype=np.float32)
    seg32 = slic.astype(np.int32)
    mean = orc.color2d_mean(img32, seg32)
    std = np.sqrt(orc.color2d_variance(img32, seg32, mean.astype(np.float32)))
    energy = orc.color2d_energy(img32, seg32)
    features = np.nan_to_num(np.hstack([mean, std, energy]))
    t2 = time.perf_counter()
    proba = model.predict_proba(features)
    _, edges = orc.adjacency(seg32)
    edges = np.array(edges, dtype=np.int32)
    centres = orc.centers(seg32)
    weights = gc.compute_edge_model(edges, proba, 'lT')
    weights = weights / gc.compute_spatial_dist([tuple(c) for c in centres], edges, relative=True)
    weights = np.clip(weights, 1e-3, 1e3)
    unary = gc.compute_unary_cost(proba)
    pairwise = gc.compute_pairwise_cost(GC_REGUL, proba.shape)
    t3 = time.perf_counter()
    labels = orc.cut_general_graph(edges, weights, unary, pairwise, n_iter=-1)
    t4 = time.perf_counter()
    segm = labels[slic]
    soft = proba[slic]
    t5 = time.perf_counter()
    return t5 - t0, (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4), segm, soft


def main():
    args = parse_args()
    try:    # the host stages multiply 2025 x 9 matrices: one BLAS thread each, N ranks x M worker threads share the node
        from threadpoolctl import threadpool_limits
        threadpool_limits(limits=1)
    except Exception:
        pass
    from pyimsegm_amd.distributed import Group
    group = Group()                      # torch.distributed (RCCL) only when launched by torchrun
    world, rank = group.world, group.rank

    from pyimsegm_amd import _hip
    from pyimsegm_amd import pipelines as pipe
    from pyimsegm_amd.descriptors import FEATURES_SET_COLOR
    from pyimsegm_amd.graph_cuts import estim_class_model, predict_proba
    from pyimsegm_amd.superpixels import _open_session
    from pyimsegm_amd.utilities.synthetic import voronoi_image

    size = args.size
    image = voronoi_image(size, size, seed=1 + rank)
    ctx = _hip.default_context()
    sess, mode = _open_session(image)           # H2D once: the image is resident from here on

    host_pool = None

    def step(model, session, to_host=False):
        res = pipe._ResidentImage(image, FEATURES_SET_COLOR, SP_SIZE, SP_REGUL, session=session)
        if host_pool is not None:
            return res.segment_with_model(model, GC_REGUL, EDGE_TYPE, host_pool, to_host=to_host), res
        proba = predict_proba(model, res.features)
        return res.segment(proba, GC_REGUL, EDGE_TYPE, to_host=to_host), res

    # model fit once (outside the timed region): the reference's group-model flow, run_segm...:476-514
    np.random.seed(0)
    res0 = pipe._ResidentImage(image, FEATURES_SET_COLOR, SP_SIZE, SP_REGUL, session=(sess, mode))
    model = estim_class_model(res0.features, NB_CLASSES, 'GMM', None, True)

    # Worker threads keep `inflight` images of this rank in flight: each has its own context (HIP stream)
    # and its own resident copy of the image; the host stages of one step overlap the kernels of another.
    # Step i of the K timed steps is taken by the next free worker; with more than one rank every finished
    # label map is gathered on rank 0 by the main thread (one RCCL gather per step, zero copy from HBM).
    import queue
    import threading
    # measured on MI355X (total ms for K steps at 2 / 3 / 4 / 6 in flight): K=10: 21 / 19 / 18 / 24, K=20: 34 / 31 / 32 / 32,
    # K=50: 85 / 66 / 66 / 59, K=100: - / - / 129 / 104
    inflight = args.inflight if args.inflight > 0 else (6 if args.steps >= 40 else 4 if args.steps >= 8 else min(3, max(1, args.steps)))
    if args.host_pool and inflight > 1:
        # the numpy stages of the images in flight leave the interpreter lock of this process (same functions,
        # same numbers: pyimsegm_amd/hostpool.py); helpers are started and given the model before the timed region
        from pyimsegm_amd.hostpool import HostMathPool
        try:
            host_pool = HostMathPool(inflight)
            host_pool.set_model(model)
        except Exception as ex:       # no helpers (cannot spawn, model not picklable ...): the threads do the numpy work
            print('host helper processes not available (%r): numpy stages stay in the worker threads' % (ex, ), file=sys.stderr)
            if host_pool is not None:
                host_pool.close()
            host_pool = None
    if args.switch_interval:
        sys.setswitchinterval(args.switch_interval)
    do_gather = group.dist is not None            # launched by torchrun (also exercised with a single rank)
    todo = queue.Queue()
    done = queue.Queue()
    ready = threading.Barrier(inflight + 1)
    contexts = [None] * inflight

    errors = []

    def worker(idx):
        try:
            wctx = _hip.default_context()            # per thread
            contexts[idx] = wctx
            session = _open_session(image)
            for _ in range(max(args.warmup, 1)):
                step(model, session)
            wctx.synchronize()
            ready.wait()
            while True:
                item = todo.get()
                if item is None:
                    break
                step(model, session)
                if do_gather:
                    gathered = threading.Event()
                    done.put((session[0], gathered))
                    gathered.wait()                   # the label buffer is reused by the next step
                else:
                    done.put((session[0], None))
            wctx.synchronize()
        except BaseException as ex:                   # never leave the main thread waiting for a dead worker
            errors.append(ex)
            ready.abort()
        finally:
            done.put(None)

    threads = [threading.Thread(target=worker, args=(i,), daemon=True) for i in range(inflight)]
    for t in threads:
        t.start()
    try:
        ready.wait()                              # sessions resident, warm-up done
    except threading.BrokenBarrierError:
        raise RuntimeError('a worker thread failed during warm-up: %r' % (errors[:1], ))
    group.barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        todo.put(i)
    for _ in threads:
        todo.put(None)
    finished = 0
    while finished < inflight:
        item = done.get()
        if item is None:
            finished += 1
            continue
        if do_gather:
            item[0].ctx.synchronize()
            group.gather_arrays(_hip.segm_device_array(item[0]), dst=0, keep_on_device=True)
            item[1].set()
    group.barrier()
    elapsed = time.perf_counter() - t0
    for t in threads:
        t.join()
    if errors:
        raise RuntimeError('a worker thread failed: %r' % (errors[0], ))
    elapsed = group.max_over_ranks(elapsed)

    # Kernel-level figures (roofline of the dominant kernel, stage breakdown): a second, un-overlapped pass of
    # a few steps on one stream with HIP events around every stage -- with several images in flight the
    # kernels of different streams share the GPU and their individual durations say nothing.
    prof_steps = min(max(args.steps, 1), 5)
    ctx.profile_enable(True)
    ctx.profile_reset()
    for _ in range(prof_steps):
        step(model, (sess, mode))
    ctx.synchronize()
    assign_ms, assign_n = ctx.profile_get('slic_assign')
    stage_ms = {g: ctx.profile_get(g) for g in _hip.PROFILE_GROUPS}
    ctx.profile_enable(False)

    if rank == 0:
        npx = size * size
        value = world * args.steps * npx / elapsed / 1e6
        avg_assign_s = assign_ms / max(assign_n, 1) / 1e3
        achieved = ASSIGN_BYTES_PER_PX * npx / avg_assign_s / 1e9 if assign_n else 0.0
        traffic = None
        try:   # HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes
            with open(os.path.join(ROOT, 'profiles', 'pmc_traffic.json')) as fp:
                pmc = json.load(fp)
            if size == HEIGHT:
                traffic = pmc['hbm_bytes_per_launch']
        except Exception:
            pass
        out = {
            'metric': 'Mpixels/s end-to-end SLIC+fts+GC, 2048x2048 RGB; % HBM roofline',
            'value': round(value, 3),
            'unit': 'Mpixels/s',
            'n_gpus': world,
            'steps': args.steps,
            'warmup': args.warmup,
            'ms_per_step': round(elapsed / args.steps * 1e3, 4),
            'higher_is_better': True,
            'scaling': 'weak',
            'vs_baseline': None,
            'dtype': 'f64',
            'data': 'synthetic',
            'config': {
                'workload': 'single %dx%d RGB uint8 per GPU, SLIC(sp_size=46, n_segments=1982, K=2025, 10 sweeps) + '
                            'colour mean/std/energy + 3-class alpha-expansion GC (gc_regul=2.0, edge=model), '
                            'pre-fitted GMM (BASELINE configs[1])' % (size, size),
                'images_per_step_per_gpu': 1,
                'images_in_flight_per_gpu': inflight,
                'host_math': ('%d helper processes per GPU (scikit-learn class model + graph-cut terms)' % inflight)
                if host_pool is not None else 'in the worker threads',
                'parallelism': 'images sharded over %d GPU(s), %d in flight per GPU (one HIP stream each), RCCL gather '
                               'of label maps' % (world, inflight),
            },
            'roofline': {
                'bound': 'hbm',
                'kernel': 'k_slic_assign_dot (assignment + fused centroid accumulation; all sweeps of the pass)',
                'achieved': round(achieved, 2),
                'peak': HBM_PEAK_GBS,
                'unit': 'GB/s',
                'frac': round(achieved / HBM_PEAK_GBS, 5),
                'traffic': traffic,
                'traffic_source': 'profiles/pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes)',
                'avg_kernel_us': round(avg_assign_s * 1e6, 3),
                'launches': assign_n,
                'algorithmic_bytes_per_launch': ASSIGN_BYTES_PER_PX * npx,
            },
            'stage_ms_per_step': {g: round(ms / prof_steps, 4) for g, (ms, n) in stage_ms.items()},
            'stage_note': 'stage and roofline figures: separate pass of %d un-overlapped steps on one stream' % prof_steps,
        }
        if world == 1 and not args.no_cpu_baseline:
            try:
                base, segm_cpu, _ = cpu_baseline(image, model)
                out['cpu_baseline'] = base
                (segm_gpu, _), res_gpu = step(model, (sess, mode), to_host=True)
                out['gpu_equals_cpu_oracle'] = bool(np.array_equal(segm_gpu, segm_cpu))
                out['speedup_vs_cpu_baseline'] = round(value / base['value'], 2)
                try:      # the third-party reference itself, when the box has it (an extra: never loses the line)
                    real = real_skimage_slic(image)
                    if real is not None:
                        base['reference_slic'] = {'scikit_image': real[0], 'seconds': round(real[1], 3), 'cores': 1,
                                                  'note': 'skimage.segmentation.slic as called at imsegm/superpixels.py:'
                                                          '61-63, same image; the oracle SLIC time is in `sample`'}
                        out['gpu_slic_equals_scikit_image'] = bool(np.array_equal(res_gpu.slic, real[2]))
                except Exception:
                    pass
            except Exception as ex:   # the baseline is a reported extra, never a reason to lose the line
                out['cpu_baseline'] = {'error': repr(ex)}
        if world == 1 and size == HEIGHT:
            try:      # the reference's own run on this very image (build container, tests/golden/make_golden_reference.py)
                verdict = compare_with_reference_run(image, sess, mode, pipe, predict_proba)
                if verdict is not None:
                    out.update(verdict)
            except Exception:
                pass
        print(json.dumps(out), flush=True)
    if host_pool is not None:
        host_pool.close()
    group.close()


if __name__ == '__main__':
    main()
