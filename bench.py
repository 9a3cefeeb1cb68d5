#!/usr/bin/env python
"""bench.py -- headline benchmark of the SLIC -> descriptors -> GraphCut hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config 2|3|4|5] [--inflight M]

What one STEP is ("image resident in host numpy" -> "`segm` resident in host numpy"):

  config 2 (default, BASELINE configs[1], the configuration the metric is quoted on)
      one synthetic 2048 x 2048 RGB uint8 image per GPU through `segment_color2d_slic_features_model_graphcut`
      (imsegm/pipelines.py:160): H2D of the image, SLIC (sp_size 46 -> K = 2025 centroids, 10 sweeps, connectivity),
      colour mean / std / energy, the pre-fitted 3-class mixture (evaluated on the device), unary / edge terms,
      alpha-expansion (gc_regul 2.0, edge type 'model'), `classes_[graph_labels][slic]`, D2H of `segm` (int32).
      The class model is fitted once during warm-up (the reference's group-model flow, run_segm...:476-514).
  config 3  same image, `{'tLM': ('mean', 'std', 'energy')}` (Leung-Malik bank, F = 180): `--config 3`
  config 4  a batch of 8 images of 647 x 1024 RGB uint8 per GPU (BASELINE: 64 images over 8 GPUs, seeds 100..163) through
            the one-call pipeline with the GROUP model of the reference's own run over the 64 images: `--config 4`
  config 5  one 64 x 4096 x 4096 float32 volume through `pipe_gray3d_slic_features_model_graphcut` (model fit
            included, as the reference function does): `--config 5`

Without `--config` the line of config 2 carries `other_configs`: configs 3, 4 and a reduced config 5 (the 32 x 512 x 512 volume
the reference itself was run on) timed in the same invocation with a few steps each, every one with its own equality keys
against the reference's run at that size (tests/golden/reference_c{3,4,5}.npz, tests/golden/make_golden_configs.py).

`value` = pixels (voxels) of all timed steps of all ranks / wall time, host to host.  M images are kept in flight per
GPU (worker threads with one HIP stream and one recycled session each; a thread spends a step inside a handful of C
calls, outside the interpreter lock) -- the reference maps a pool of worker processes over the images.  Timing is
taken in STEADY STATE: W + K + M steps are issued back to back between two barriers + device synchronisations, and
the clock runs from the completion of step W to the completion of step W + K, so exactly K steps are timed and the
figure does not depend on the fill and drain of the pipeline (`ms_per_step_incl_fill_drain` is reported next to it).
Separate keys carry the device-resident rate of the same pipeline (image already in HBM, result left in HBM:
`device_resident`), the extra cost of fetching `segm_soft` (`soft_d2h_ms`), and a single image end to end with
nothing else in flight (`latency_ms`).

`roofline` and `stage_ms_per_step` come from a separate, un-overlapped pass on one stream (the dominant kernel is timed
by a HIP event pair attached to its dispatch).  N > 1: one process per GPU -- ranks from the launcher's environment, or,
when `--gpus N` is given without one, N ranks spawned by this script itself -- every rank segments its own image(s) (weak
scaling, no data-path collective) and the label maps of every round of M steps are gathered in rank 0's HBM with one
grouped RCCL send / recv (pyimsegm_amd.distributed, ctypes on librccl); rank 0 checks the CRC of EVERY gathered map of a
verification round against the reference's run.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import threading
import time
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')

SP_SIZE, SP_REGUL, NB_CLASSES, GC_REGUL, EDGE_TYPE = 46, 0.2, 3, 2.0, 'model'
HEIGHT = WIDTH = 2048
C4_SHAPE, C4_SP_SIZE, C4_PER_STEP, C4_FIRST_SEED, C4_NB_IMAGES = (647, 1024), 35, 8, 100, 64   # run_segm...:105-106 slic_size 35
C5_PARAMS = dict(spacing=(1, 1, 1), sp_size=15, sp_regul=0.2, gc_regul=0.1)
C5_REDUCED = (32, 512, 512)
C5_FULL = (64, 4096, 4096)
FEATURES_LM = {'tLM': ('mean', 'std', 'energy')}
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s
FP64_VALU_PEAK_TFLOPS = 78.6   # MI355X_MICROARCH.md: fp64 vector (non-MFMA) peak
ASSIGN_BYTES_PER_PX = 28.0     # read fp64 Lab 3 x 8 B + write int32 label 4 B per pixel per sweep
VOL_ASSIGN_BYTES_PER_VOXEL = 8.0   # float32 volume: read 4 B + write int32 label 4 B per voxel per sweep
METRIC = 'Mpixels/s end-to-end SLIC+fts+GC, 2048x2048 RGB; % HBM roofline'


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=None)
    ap.add_argument('--steps', type=int, default=None)
    ap.add_argument('--warmup', type=int, default=None)
    ap.add_argument('--config', type=int, default=None, choices=(2, 3, 4, 5),
                    help='default: config 2 with configs 3, 4 and a reduced 5 appended as `other_configs`')
    ap.add_argument('--size', type=int, default=None, help='image edge of configs 2 / 3 (default: the BASELINE 2048)')
    ap.add_argument('--volume', type=str, default='64,4096,4096', help='D,H,W of config 5')
    ap.add_argument('--inflight', type=int, default=0, help='images in flight per GPU (worker threads); 0 = default of the config')
    ap.add_argument('--pinned-input', type=int, default=0, help='1: the input images live in page-locked host memory')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-other-configs', action='store_true')
    ap.add_argument('--batch-images', type=int, default=-1, help='config 4: 0 = one image per launch chain (round 3), default: the 8 '
                                                                 'images of a step in ONE launch chain (imsegm_batch2d_run_color)')
    ap.add_argument('--input-ring', type=int, default=-1, help='1: every worker thread copies the images of a step into its own page-locked '
                                                               'buffers first (one memcpy, then DMA) instead of handing pageable arrays to the '
                                                               'runtime (default off: measured slower on one GPU)')
    ap.add_argument('--probe-seconds', type=float, default=1.0, help='N > 1, --input-ring -1: length of each of the two input-staging probes')
    ap.add_argument('--no-full-volume', action='store_true', help='leave BASELINE configs[4] at its full 64x4096x4096 out of `other_configs`')
    return ap.parse_args(argv)


def crc32(arr, dtype=np.int32):
    return zlib.crc32(np.ascontiguousarray(arr, dtype=dtype).tobytes())


def load_golden(name):
    path = os.path.join(GOLDEN, name)
    return np.load(path, allow_pickle=False) if os.path.exists(path) else None


def model_from_arrays(ref):
    """scikit-learn `Pipeline([StandardScaler, GaussianMixture('full')])` with the parameters of a run of the reference
    (tests/golden/reference_*.npz)"""
    from sklearn.mixture import GaussianMixture
    from sklearn.pipeline import Pipeline
    from sklearn.preprocessing import StandardScaler
    scaler = StandardScaler()
    scaler.mean_, scaler.scale_ = ref['scaler_mean'], ref['scaler_scale']
    scaler.var_, scaler.n_features_in_ = scaler.scale_**2, len(scaler.mean_)
    scaler.n_samples_seen_ = 1
    gmm = GaussianMixture(n_components=len(ref['gmm_weights']), covariance_type='full')
    gmm.weights_, gmm.means_ = ref['gmm_weights'], ref['gmm_means']
    gmm.covariances_, gmm.precisions_cholesky_ = ref['gmm_covariances'], ref['gmm_precisions_cholesky']
    gmm.precisions_ = np.array([pc @ pc.T for pc in gmm.precisions_cholesky_])
    gmm.converged_, gmm.n_features_in_ = True, scaler.n_features_in_
    return Pipeline([('scaler', scaler), ('GMM', gmm)])


def median_min(values):
    values = sorted(values)
    return values[len(values) // 2], values[0]


# -----------------------------------------------------------------------------------------------------------------
# CPU baseline (rank 0, N = 1): the reference's own legs where they exist on the box, the oracle port for the rest.
# warm-up 1, median of 5.
# -----------------------------------------------------------------------------------------------------------------
_SKIMAGE_SCRIPT = r'''
import sys, time, warnings
warnings.filterwarnings("ignore")
import numpy as np
from skimage.segmentation import slic
import skimage
img = np.load(sys.argv[1])
sp_size, rc, repeats = float(sys.argv[3]), float(sys.argv[4]), int(sys.argv[5])
nb_pixels = img.shape[0] * img.shape[1]
if img.min() != 0. or img.max() != 1.:                       # imsegm/superpixels.py:53-54
    img = (img - img.min()) / float(img.max() - img.min())
times = []
for _ in range(repeats):
    t0 = time.perf_counter()
    labels = slic(img, n_segments=int(nb_pixels / sp_size**2), compactness=(sp_size * rc)**1.5, sigma=1,
                  enforce_connectivity=True, slic_zero=False)      # imsegm/superpixels.py:57-63
    times.append(time.perf_counter() - t0)
np.save(sys.argv[2], np.asarray(labels).astype(np.int32))
print("%s %s" % (skimage.__version__, ",".join("%.4f" % t for t in times)))
'''


def real_skimage_slic(image, sp_size, sp_regul, repeats=1):
    """the REAL `skimage.segmentation.slic` behind imsegm/superpixels.py:61-63, when the box carries the image's conda
    Python 3.9 with scikit-image (the interpreter of this script cannot import it): (version, seconds of every one of
    `repeats` slic calls on one host core, label map), or None"""
    import subprocess
    import tempfile
    py = os.environ.get('IMSEGM_SKIMAGE_PYTHON', '/opt/conda/bin/python3.9')
    if not os.path.exists(py):
        return None
    try:
        with tempfile.TemporaryDirectory() as tmp:
            src, dst = os.path.join(tmp, 'image.npy'), os.path.join(tmp, 'labels.npy')
            np.save(src, image)
            run = subprocess.run([py, '-c', _SKIMAGE_SCRIPT, src, dst, str(sp_size), str(sp_regul), str(repeats)],
                                 capture_output=True, text=True, timeout=900)
            if run.returncode != 0:
                return None
            version, seconds = run.stdout.split()[-2:]
            return version, [float(s) for s in seconds.split(',')], np.load(dst)
    except Exception:
        return None


_SKIMAGE3D_SCRIPT = r'''
import sys, time, warnings
warnings.filterwarnings("ignore")
import numpy as np
from skimage import measure
from skimage.segmentation import slic
import skimage
im = np.load(sys.argv[1])
sp_size, rc = float(sys.argv[3]), float(sys.argv[4])
space = tuple(float(v) for v in sys.argv[5].split(","))
nb_pixels = np.prod(im.shape)                                          # imsegm/superpixels.py:93-97
sp = np.prod(sp_size / np.asarray(space, dtype=np.float32) * min(space))
t0 = time.perf_counter()
seg = slic(np.array(im), n_segments=int(nb_pixels / sp), compactness=int((sp * rc)**1.5), multichannel=False, spacing=space, sigma=1)
t1 = time.perf_counter()
seg = measure.label(seg)                                               # imsegm/superpixels.py:104-111
t2 = time.perf_counter()
np.save(sys.argv[2], np.asarray(seg).astype(np.int32))
print("%s %.4f,%.4f" % (skimage.__version__, t1 - t0, t2 - t1))
'''


def real_skimage_slic3d(vol, sp_size, sp_regul, spacing):
    """the REAL `skimage.segmentation.slic` + `skimage.measure.label` behind imsegm/superpixels.py:104-111 on a gray volume, one host
    core, under the image's conda Python 3.9 (as real_skimage_slic): (version, [slic seconds, label seconds], label map) or None"""
    import subprocess
    import tempfile
    py = os.environ.get('IMSEGM_SKIMAGE_PYTHON', '/opt/conda/bin/python3.9')
    if not os.path.exists(py):
        return None
    try:
        with tempfile.TemporaryDirectory() as tmp:
            src, dst = os.path.join(tmp, 'volume.npy'), os.path.join(tmp, 'labels.npy')
            np.save(src, vol)
            run = subprocess.run([py, '-c', _SKIMAGE3D_SCRIPT, src, dst, str(sp_size), str(sp_regul), ','.join(str(v) for v in spacing)],
                                 capture_output=True, text=True, timeout=900, env=dict(os.environ, OMP_NUM_THREADS='1'))
            if run.returncode != 0:
                return None
            version, seconds = run.stdout.split()[-2:]
            return version, [float(v) for v in seconds.split(',')], np.load(dst)
    except Exception:
        return None


def _cpu_chain_color2d(image, model, sp_size, sp_regul, gc_regul, slic=None, keep=None):
    """the CPU path of one image after (or including) SLIC on one core: descriptors = the reference's own
    features_cython.pyx compiled into oracle/_ref (else the oracle's C restatement), graph / edge weights / GraphCut /
    gathers = the oracle port (gco exists nowhere); returns (legs in seconds, segmentation, label map)"""
    from oracle import oracle as orc
    from pyimsegm_amd import graph_cuts as gc
    parts = {}
    if slic is None:
        t0 = time.perf_counter()
        slic = orc.segment_slic_img2d(image, sp_size, sp_regul)
        parts['slic_port_s'] = time.perf_counter() - t0
    img32 = np.asarray(image, dtype=np.float32)
    seg32 = slic.astype(np.int32)
    ref = orc.ref_features_cython()
    t1 = time.perf_counter()
    if ref is not None:        # descriptors.py:209-296: the reference's Cython functions on float32 image / int32 labels
        mean = np.array(ref.computeColorImage2dMean(img32, seg32))
        std = np.sqrt(np.array(ref.computeColorImage2dVariance(img32, seg32, np.array(mean, dtype=np.float32))))
        energy = np.array(ref.computeColorImage2dEnergy(img32, seg32))
        parts['descriptors_reference_s'] = time.perf_counter() - t1
    t1 = time.perf_counter()
    mean = orc.color2d_mean(img32, seg32)
    std = np.sqrt(orc.color2d_variance(img32, seg32, mean.astype(np.float32)))
    energy = orc.color2d_energy(img32, seg32)
    features = np.nan_to_num(np.hstack([mean, std, energy]))
    parts['descriptors_port_s'] = time.perf_counter() - t1
    if keep is not None:
        keep['features'] = features
    t2 = time.perf_counter()
    proba = model.predict_proba(features)
    _, edges = orc.adjacency(seg32)
    edges = np.array(edges, dtype=np.int32)
    centres = orc.centers(seg32)
    weights = gc.compute_edge_model(edges, proba, 'lT')
    weights = weights / gc.compute_spatial_dist([tuple(c) for c in centres], edges, relative=True)
    weights = np.clip(weights, 1e-3, 1e3)
    unary = gc.compute_unary_cost(proba)
    pairwise = gc.compute_pairwise_cost(gc_regul, proba.shape)
    parts['graph_terms_port_s'] = time.perf_counter() - t2
    t3 = time.perf_counter()
    labels = orc.cut_general_graph(edges, weights, unary, pairwise, n_iter=-1)
    parts['graphcut_port_s'] = time.perf_counter() - t3
    t4 = time.perf_counter()
    segm = labels[slic]
    _ = proba[slic]
    parts['gathers_s'] = time.perf_counter() - t4
    return parts, segm, slic


def cpu_baseline_color2d(image, model, sp_size, sp_regul, repeats=5):
    """one full image through the CPU path on ONE host core (the reference is single-threaded per image): 1 warm-up +
    `repeats` timed passes, median (and minimum) of the per-pass totals.  SLIC = the real scikit-image when the box has it
    (else the oracle's C restatement).  Returns (entry for the JSON line, segmentation of the port chain, label map of
    the real scikit-image or None)."""
    from oracle import oracle as orc
    orc.lib()
    real = real_skimage_slic(image, sp_size, sp_regul, repeats=repeats + 1)
    runs = []
    segm = None
    keep = {}
    for i in range(repeats + 1):
        parts, segm, _ = _cpu_chain_color2d(image, model, sp_size, sp_regul, GC_REGUL, keep=keep)
        if real is not None:
            parts['slic_reference_s'] = real[1][i]
        runs.append(parts)
    runs = runs[1:]                                   # the first pass is the warm-up

    def total(p):
        return (p.get('slic_reference_s', p['slic_port_s']) + p.get('descriptors_reference_s', p['descriptors_port_s'])
                + p['graph_terms_port_s'] + p['graphcut_port_s'] + p['gathers_s'])

    med, best = median_min([total(p) for p in runs])
    legs = {k: round(median_min([p[k] for p in runs])[0], 4) for k in runs[0]}
    npx = image.shape[0] * image.shape[1]
    kind = 'reference+port' if (real is not None or 'descriptors_reference_s' in legs) else 'port'
    entry = {
        'value': round(npx / med / 1e6, 4), 'unit': 'Mpixels/s', 'cores': 1, 'kind': kind,
        'value_best_of_%d' % repeats: round(npx / best / 1e6, 4),
        'sample': 'one full %dx%d image, 1 warm-up + %d timed passes on one core, median %.2f s (min %.2f s): SLIC %.2f s (%s), '
                  'descriptors %.3f s (%s), graph + edge weights %.2f s (port), GraphCut %.3f s (port; gco exists nowhere), '
                  'gathers %.2f s'
                  % (image.shape[0], image.shape[1], repeats, med, best, legs.get('slic_reference_s', legs['slic_port_s']),
                     'real scikit-image %s' % real[0] if real is not None else 'oracle C restatement',
                     legs.get('descriptors_reference_s', legs['descriptors_port_s']),
                     "the reference's features_cython.pyx (oracle/_ref)" if 'descriptors_reference_s' in legs else 'oracle C restatement',
                     legs['graph_terms_port_s'], legs['graphcut_port_s'], legs['gathers_s']),
        'legs_s_median': legs,
    }
    if real is not None:
        entry['reference_slic'] = {'scikit_image': real[0], 'seconds_median': round(median_min(real[1][1:])[0], 3), 'cores': 1}
    try:
        # the fit-inclusive form (the reference's per-image mode fits the mixture inside the call, pipelines.py:95): the SAME
        # scikit-learn call as on the GPU side, on the same core
        from pyimsegm_amd.graph_cuts import estim_class_model
        fits = []
        for _ in range(3):
            np.random.seed(0)
            t0 = time.perf_counter()
            estim_class_model(keep['features'], NB_CLASSES, 'GMM', None, True)
            fits.append(time.perf_counter() - t0)
        fit_s = median_min(fits)[0]
        entry['model_fit_s'] = round(fit_s, 4)
        entry['value_including_fit'] = round(npx / (med + fit_s) / 1e6, 4)
    except Exception as ex:
        entry['model_fit_error'] = repr(ex)
    try:     # the reference ITSELF, whole pipeline, timed in the build container (tools/time_reference.py)
        with open(os.path.join(ROOT, 'profiles', 'reference_time_r02.json')) as fp:
            entry['reference_whole_pipeline_build_container'] = json.load(fp)
    except Exception:
        pass
    return entry, segm, (real[2] if real is not None else None)


def cpu_baseline_texture(image, sp_size, sp_regul, budget_s=20.0):
    """config 3 on one host core of THIS box: the reference's Leung-Malik leg is scipy.ndimage.convolve per kernel and colour
    channel (/root/reference/imsegm/descriptors.py:951-966, compute_img_filter_response2d) -- a bounded sample of those
    convolutions is timed here with the installed scipy (the first kernels of the bank on the benchmark image's channels, about
    `budget_s` seconds) and multiplied up to the 76 x 3 convolutions of the bank; the sigma = 150 high-pass is timed whole; SLIC
    as in config 2.  Beside it: the seconds the reference ITSELF took for `compute_color2d_superpixels_features` on this very
    image in the build container (tests/golden/reference_c3.npz, 827 s on one core)."""
    from scipy import ndimage
    from pyimsegm_amd import descriptors as d
    filters, _ = d.create_filter_bank_lm_2d()
    kernels = [k for battery in filters for k in battery]
    img = np.asarray(image)
    t0 = time.perf_counter()
    high = img - ndimage.gaussian_filter(img, 150)                   # descriptors.py:1078 (scalar sigma: the channel axis too)
    t_blur = time.perf_counter() - t0
    planes = [np.ascontiguousarray(high[:, :, c]) for c in range(3)]
    spent, done = 0.0, 0
    for kernel in kernels:
        for plane in planes:
            t = time.perf_counter()
            ndimage.convolve(plane, kernel)                           # descriptors.py:960-963
            spent += time.perf_counter() - t
            done += 1
            if spent >= budget_s:
                break
        if spent >= budget_s:
            break
    per_conv = spent / done
    total_conv = len(kernels) * 3
    real = real_skimage_slic(img, sp_size, sp_regul, repeats=1)
    t_slic = float(np.median(real[1])) if real is not None else None
    slic_how = 'real scikit-image %s' % real[0] if real is not None else 'oracle port'
    if t_slic is None:
        from oracle import oracle as orc
        t = time.perf_counter()
        orc.segment_slic_img2d(img, sp_size, sp_regul)
        t_slic = time.perf_counter() - t
    total = t_slic + t_blur + per_conv * total_conv
    out = {'value': round(img.shape[0] * img.shape[1] / total / 1e6, 5), 'unit': 'Mpixels/s', 'cores': 1, 'kind': 'reference+port',
           'sample': 'SLIC %.2f s (%s) + sigma=150 high-pass %.1f s (scipy, whole) + %d of the %d scipy.ndimage.convolve calls of the '
                     'bank (76 kernels 33x33 x 3 channels) timed: %.2f s each, EXTRAPOLATED to %.0f s; per-superpixel statistics, '
                     'class model and GraphCut (under 0.3 s in config 2) not included'
                     % (t_slic, slic_how, t_blur, done, total_conv, per_conv, per_conv * total_conv),
           'extrapolated_seconds_per_image': round(total, 1)}
    ref = load_golden('reference_c3.npz')
    if ref is not None and 'seconds_slic_and_descriptors_one_core' in ref.files:
        sec = float(ref['seconds_slic_and_descriptors_one_core'])
        out['reference_whole_stage_build_container'] = {
            'what': 'the reference itself (unchanged, %s): compute_color2d_superpixels_features with the full Leung-Malik bank on this '
                    'image, one core of the build container (NOT the GPU box)' % str(ref['versions']),
            'seconds': round(sec, 1), 'mpixels_per_s': round(img.shape[0] * img.shape[1] / sec / 1e6, 5)}
    return out


def _pool_one_image(task):
    """one config-4 image through the CPU chain on one core; SLIC = the REAL scikit-image when this interpreter carries it (the
    pool then runs under the image's conda Python, see cpu_baseline_batch), else the oracle port.  Returns (seconds, skimage
    version or None)."""
    seed, sp_size, sp_regul, gc_regul, arrays = task
    try:
        from threadpoolctl import threadpool_limits
        threadpool_limits(limits=1)
    except Exception:
        pass
    from pyimsegm_amd.utilities.synthetic import voronoi_image
    image = voronoi_image(C4_SHAPE[0], C4_SHAPE[1], seed=seed)
    version = None
    t0 = time.perf_counter()
    slic = None
    try:
        import warnings
        import skimage
        from skimage.segmentation import slic as sk_slic
        img = np.asarray(image, dtype=np.float64)
        if img.min() != 0. or img.max() != 1.:                       # imsegm/superpixels.py:53-54
            img = (img - img.min()) / float(img.max() - img.min())
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            slic = np.asarray(sk_slic(img, n_segments=int(img.shape[0] * img.shape[1] / float(sp_size)**2),
                                      compactness=(sp_size * sp_regul)**1.5, sigma=1, enforce_connectivity=True, slic_zero=False))
        version = skimage.__version__
    except ImportError:
        pass
    _cpu_chain_color2d(image, model_from_arrays(arrays), sp_size, sp_regul, gc_regul, slic=slic)
    return time.perf_counter() - t0, version


def _pool_batch(model_arrays, sp_size, sp_regul, nb_images=None):
    import multiprocessing as mp
    nproc = os.cpu_count() or 1
    workers = max(1, int(0.9 * nproc))
    nb_images = nb_images or min(C4_NB_IMAGES, 2 * workers)
    tasks = [(C4_FIRST_SEED + i, sp_size, sp_regul, GC_REGUL, model_arrays) for i in range(nb_images)]
    with mp.get_context('spawn').Pool(workers) as pool:
        pool.map(_pool_one_image, tasks[:workers])                  # warm-up: imports, library loads
        t0 = time.perf_counter()
        got = pool.map(_pool_one_image, tasks, chunksize=1)
        wall = time.perf_counter() - t0
    per_image, versions = [g[0] for g in got], sorted({g[1] for g in got if g[1]})
    npx = C4_SHAPE[0] * C4_SHAPE[1]
    real = versions[0] if (versions and all(g[1] for g in got)) else None
    return {'value': round(nb_images * npx / wall / 1e6, 4), 'unit': 'Mpixels/s', 'cores': workers,
            'kind': 'reference+port' if real else 'port',
            'sample': '%d images of %dx%d over a pool of %d worker processes (int(0.9 * %d cpus), as the reference driver), wall %.2f s; '
                      'one image on one core: median %.2f s; SLIC = %s, the other legs = the oracle port'
                      % (nb_images, C4_SHAPE[0], C4_SHAPE[1], workers, nproc, wall, median_min(per_image)[0],
                         'the real scikit-image %s' % real if real else 'the oracle C restatement'),
            'one_core_value': round(npx / median_min(per_image)[0] / 1e6, 4)}


def cpu_baseline_batch(model_arrays, sp_size, sp_regul, nb_images=None):
    """config 4 on the host as the reference's driver runs it: a pool of `int(0.9 * nproc)` worker processes mapped over
    the images (run_segm_slic_model_graphcut.py:61, experiments.py:392-403), every worker one image at a time on one core;
    bounded sample of 2 images per worker.  SLIC -- 97 % of the chain -- is the REAL scikit-image: the pool runs under the
    image's conda Python 3.9 when the box carries it (this interpreter cannot import scikit-image); otherwise the oracle port,
    labelled so."""
    try:
        import skimage  # noqa: F401
        return _pool_batch(model_arrays, sp_size, sp_regul, nb_images)
    except ImportError:
        pass
    py = os.environ.get('IMSEGM_SKIMAGE_PYTHON', '/opt/conda/bin/python3.9')
    if os.path.exists(py):
        import subprocess
        import tempfile
        try:
            with tempfile.TemporaryDirectory() as tmp:
                path = os.path.join(tmp, 'model.npz')
                np.savez(path, **model_arrays)
                code = ('import sys, json, warnings; warnings.filterwarnings("ignore"); sys.path.insert(0, %r); import numpy as np; import bench; '
                        'arrays = dict(np.load(sys.argv[1])); '
                        'print("POOL " + json.dumps(bench._pool_batch(arrays, float(sys.argv[2]), float(sys.argv[3]), int(sys.argv[4]) or None)))' % ROOT)
                run = subprocess.run([py, '-c', code, path, str(sp_size), str(sp_regul), str(nb_images or 0)], capture_output=True, text=True,
                                     timeout=900, env=dict(os.environ, OMP_NUM_THREADS='1'))
                lines = [ln for ln in run.stdout.splitlines() if ln.startswith('POOL ')]
                if run.returncode == 0 and lines:
                    return json.loads(lines[-1][5:])
        except Exception:
            pass
    return _pool_batch(model_arrays, sp_size, sp_regul, nb_images)


def cpu_baseline_volume(shape_full, crop=(64, 256, 256)):
    """config 5 on one host core: the oracle port (C restatement of scikit-image's float32 3-D SLIC + measure.label, the
    reference's gray statistics, graph, GraphCut) on a crop, extrapolated linearly in the number of voxels (BASELINE.md
    section 3: the reference's own per-voxel Python loops make the full volume infeasible)"""
    from oracle import oracle as orc
    from pyimsegm_amd import descriptors as d
    from pyimsegm_amd import graph_cuts as gc
    from pyimsegm_amd.utilities.synthetic import config5_volume
    crop = tuple(min(c, s) for c, s in zip(crop, shape_full))
    vol = config5_volume(crop)
    p = C5_PARAMS
    # supervoxels: the REAL scikit-image where the box carries it (the reference's own third-party leg), else the oracle port
    real = real_skimage_slic3d(vol, p['sp_size'], p['sp_regul'], p['spacing'])
    t0 = time.perf_counter()
    slic = orc.segment_slic_img3d_gray(vol, p['sp_size'], p['sp_regul'], p['spacing'])
    t_port = time.perf_counter() - t0
    t_slic = sum(real[1]) if real is not None else t_port
    t0 = time.perf_counter() - t_slic                      # (the clock of the legs below continues behind the leg that counts)
    seg32 = slic.astype(np.int32)
    mean = orc.gray3d_stat(vol, seg32, 'mean')
    var = orc.gray3d_stat(vol, seg32, 'var', mean.astype(np.float32))
    energy = orc.gray3d_stat(vol, seg32, 'energy')
    features = np.nan_to_num(np.stack([mean, np.sqrt(var), energy], axis=1))
    features, _ = d.norm_features(features)
    t1 = time.perf_counter()
    np.random.seed(0)
    model = gc.estim_class_model(features, NB_CLASSES)
    proba = model.predict_proba(features)
    t_fit = time.perf_counter() - t1
    _, edges = orc.adjacency(slic)
    edges = np.array(edges, dtype=np.int32)
    weights = gc.compute_edge_model(edges, proba, 'lT')
    weights = np.clip(weights / gc.compute_spatial_dist(orc.centers(slic), edges, relative=True), 1e-3, 1e3)
    labels = orc.cut_general_graph(edges, weights, gc.compute_unary_cost(proba), gc.compute_pairwise_cost(p['gc_regul'], proba.shape),
                                   n_iter=-1)
    _ = np.asarray(labels)[slic]
    total = time.perf_counter() - t0
    nvox = int(np.prod(crop))
    extra = {}
    full = load_golden('reference_c5_full.npz')
    if full is not None and tuple(int(v) for v in full['shape']) == tuple(shape_full):
        # the reference's own third-party leg on the full volume, timed in the build container (tests/golden/make_golden_configs.py c5full)
        sec = [float(v) for v in full['seconds_one_core']]
        extra['reference_slic_full_volume_build_container'] = {
            'what': 'real scikit-image %s slic (%.0f s) + measure.label (%.0f s) on the full volume, one core of the build container'
                    % (str(full['versions']), sec[0], sec[1]), 'mvoxels_per_s': round(float(np.prod(shape_full)) / sum(sec) / 1e6, 4)}
    if real is not None:
        extra['reference_slic'] = {'scikit_image': real[0], 'slic_seconds': round(real[1][0], 3), 'label_seconds': round(real[1][1], 3),
                                   'port_seconds': round(t_port, 3), 'port_equals_scikit_image': bool(np.array_equal(real[2], seg32))}
    return {**extra, 'value': round(nvox / total / 1e6, 4), 'unit': 'Mpixels/s', 'cores': 1, 'kind': 'reference+port' if real is not None else 'port',
            'sample': '%dx%dx%d %s on one core, %.1f s (SLIC + measure.label %.1f s [%s], mixture fit + predict_proba %.2f s -- the same '
                      'scikit-learn call as inside the GPU step --, statistics / graph / GraphCut / gather %.2f s)%s'
                      % (crop + ('crop of the volume' if crop != tuple(shape_full) else 'volume (whole)', total, t_slic,
                                 'real scikit-image ' + real[0] if real is not None else 'oracle C restatement', t_fit,
                                 total - t_slic - t_fit,
                                 '; the rate is EXTRAPOLATED linearly to the full volume' if crop != tuple(shape_full) else '')),
            'seconds': {'total': round(total, 3), 'slic_and_label': round(t_slic, 3), 'model_fit': round(t_fit, 3)},
            'extrapolated_seconds_full_volume': round(total * float(np.prod(shape_full)) / nvox, 1)}


# -----------------------------------------------------------------------------------------------------------------
# equality with the reference's own run at the size of each config (tests/golden/make_golden_*.py)
# -----------------------------------------------------------------------------------------------------------------
def compare_config2(image, pipe):
    """one extra, untimed GPU pass with the class model of the reference's own run on the benchmark image (real
    scikit-image 0.18.3 + the reference's Cython descriptors + its scikit-learn GMM, gco bridged to the oracle; label maps
    stored as CRC32): the superpixel map and the final segmentation of the GPU must have the same checksums"""
    ref = load_golden('reference_2048.npz')
    if ref is None or crc32(image, np.uint8) != int(ref['image_crc']):
        return None
    from pyimsegm_amd.descriptors import FEATURES_SET_COLOR
    model = model_from_arrays(ref)
    res = pipe._ResidentImage(image, FEATURES_SET_COLOR, SP_SIZE, SP_REGUL)
    try:
        segm, _ = res.segment(None, GC_REGUL, EDGE_TYPE, to_host=True, want_soft=False, model=model)
        slic = res.slic
    finally:
        res.close()
    slic_ok = crc32(slic) == int(ref['slic_crc'])
    segm_ok = crc32(segm) == int(ref['segm_crc'])
    return {'gpu_equals_reference_run': bool(slic_ok and segm_ok),
            'reference_run': 'tests/golden/reference_2048.npz: the reference itself on this image (%s), superpixel map %s, '
                             'segmentation %s (the cut of that run is the oracle\'s: gco exists nowhere)'
                             % (str(ref['versions']), 'equal' if slic_ok else 'DIFFERENT', 'equal' if segm_ok else 'DIFFERENT')}


def compare_config3(image, pipe):
    """configs[2] at full size: superpixel map (CRC), the K x 180 Leung-Malik descriptors (1e-5) and, under the reference
    run's class model, the segmentation (CRC) against tests/golden/reference_c3.npz"""
    ref = load_golden('reference_c3.npz')
    if ref is None or crc32(image, np.uint8) != int(ref['image_crc']):
        return None
    model = model_from_arrays(ref)
    res = pipe._ResidentImage(image, FEATURES_LM, SP_SIZE, SP_REGUL)
    try:
        features = np.array(res.features)
        segm, _ = res.segment(None, GC_REGUL, EDGE_TYPE, to_host=True, want_soft=False, model=model)
        slic = res.slic
    finally:
        res.close()
    slic_ok = crc32(slic) == int(ref['slic_crc'])
    fts_ok = features.shape == ref['features'].shape and bool(np.allclose(features, ref['features'], rtol=1e-5, atol=1e-5))
    err = float(np.max(np.abs(features - ref['features']))) if features.shape == ref['features'].shape else None
    segm_ok = crc32(segm) == int(ref['segm_crc'])
    counts = np.bincount(np.asarray(segm).ravel(), minlength=len(ref['class_counts']))
    return {'gpu_equals_reference_run': bool(slic_ok and fts_ok and segm_ok),
            'reference_run': 'tests/golden/reference_c3.npz (%s): superpixel map %s, %d x %d descriptors %s (max abs difference %.3g, '
                             'tolerance 1e-5), segmentation %s (pixels per class %s vs %s)'
                             % (str(ref['versions']), 'equal' if slic_ok else 'DIFFERENT', features.shape[0], features.shape[1],
                                'equal within tolerance' if fts_ok else 'DIFFERENT', err if err is not None else float('nan'),
                                'equal' if segm_ok else 'DIFFERENT', counts.tolist(), ref['class_counts'].tolist())}


def compare_config5(shape, pipe, vol=None):
    """configs[4]: the float32 supervoxel map against the real scikit-image's (CRC), and on the reduced volume also the
    descriptors (1e-5) and -- under the class probabilities of the reference's run -- the segmentation (CRC)"""
    from pyimsegm_amd.descriptors import compute_selected_features_gray3d
    from pyimsegm_amd.graph_cuts import compute_pairwise_cost
    from pyimsegm_amd.superpixels import _open_volume, _run_slic3d
    from pyimsegm_amd.utilities.synthetic import config5_volume
    ref = None
    for name in ('reference_c5.npz', 'reference_c5_full.npz'):
        cand = load_golden(name)
        if cand is not None and tuple(int(v) for v in cand['shape']) == tuple(shape):
            ref = cand
    if ref is None:
        return None
    if vol is None:
        vol = config5_volume(tuple(shape))
    if crc32(vol, np.float32) != int(ref['volume_crc']):
        return None
    p = C5_PARAMS
    sess = _open_volume(vol, reuse=True)      # (the session the timed steps left: 75 GB of allocations at full size)
    try:
        _run_slic3d(sess, p['sp_size'], p['sp_regul'], p['spacing'])
        labels = sess.get_labels_int32()
        slic_ok = crc32(labels) == int(ref['slic_crc'])
        nb_ok = sess.n_labels == int(ref['nb_supervoxels'])
        out = {'gpu_slic_equals_scikit_image': bool(slic_ok and nb_ok)}
        text = 'supervoxel map %s (K = %d vs %d)' % ('equal' if slic_ok else 'DIFFERENT', sess.n_labels, int(ref['nb_supervoxels']))
        if 'features' in ref.files:
            features, _ = compute_selected_features_gray3d(vol, pipe._ShapeOnly(sess.shape), {'color': ('mean', 'std', 'energy')}, sess=sess)
            features[np.isnan(features)] = 0
            fts_ok = features.shape == ref['features'].shape and bool(np.allclose(features, ref['features'], rtol=1e-5, atol=1e-5))
            segm = sess.segment(compute_pairwise_cost(p['gc_regul'], ref['proba'].shape), 'model', proba=ref['proba'], pinned=False)['segm']
            segm_ok = crc32(segm) == int(ref['segm_crc'])
            out['gpu_equals_reference_run'] = bool(slic_ok and nb_ok and fts_ok and segm_ok)
            text += ', descriptors %s, segmentation under the class probabilities of that run %s' % (
                'equal within 1e-5' if fts_ok else 'DIFFERENT', 'equal' if segm_ok else 'DIFFERENT')
        out['reference_run'] = '%s (%s): %s' % ('tests/golden/reference_c5.npz' if 'features' in ref.files else
                                                'tests/golden/reference_c5_full.npz', str(ref['versions']), text)
        return out
    finally:
        sess.close()


# -----------------------------------------------------------------------------------------------------------------
# steady-state runner: M worker threads take W + K + M steps back to back
# -----------------------------------------------------------------------------------------------------------------
RING_ROUNDS = 4


class SteadyRun(object):
    def __init__(self, group, inflight, make_worker_state, do_step, gather_item_bytes=0, items_per_step=1):
        """make_worker_state() -> per-thread state (own HIP context); do_step(state, index, stage) runs one step and
        calls stage(k, handle) for the label map of its k-th image (handle: device array of the map); with more than
        one rank the maps of every round of `inflight` steps go to rank 0 in one grouped RCCL call.  No host-side exchange
        happens inside the timed region: a rank whose worker failed keeps taking part in the rounds (with whatever its send
        ring holds) and the error is agreed on after the last round."""
        self.group, self.inflight = group, inflight
        self.make_worker_state, self.do_step = make_worker_state, do_step
        self.gather = None
        self.item_bytes, self.items_per_step = gather_item_bytes, items_per_step
        if group.distributed and gather_item_bytes:
            from pyimsegm_amd.distributed import DeviceGather
            self.gather = DeviceGather(group, gather_item_bytes * items_per_step, inflight, depth=RING_ROUNDS)

    def run(self, warmup, steps):
        from pyimsegm_amd import _hip
        group, M = self.group, self.inflight
        total = warmup + steps + M                # M cool-down steps keep the pipeline full while the last timed steps finish
        rounds = (total + M - 1) // M
        done = [None] * total                     # completion time of every step
        staged = [threading.Event() for _ in range(total)]
        lock = threading.Lock()
        cond = threading.Condition()
        nxt = [0]
        flushed = [0]
        errors = []
        ready = threading.Barrier(M + 1)
        go = threading.Event()

        def worker():
            try:
                state = self.make_worker_state()
                self.do_step(state, -1, lambda k, handle: None)     # first-use allocations of this thread's session
                ready.wait()
                go.wait()
                while True:
                    with lock:
                        i = nxt[0]
                        nxt[0] += 1
                    if i >= total:
                        break
                    rnd, item = divmod(i, M)
                    if self.gather is not None:
                        with cond:                                    # the send ring holds RING_ROUNDS rounds
                            cond.wait_for(lambda: flushed[0] >= rnd - (RING_ROUNDS - 1) or errors)

                        def stage(k, handle, rnd=rnd, item=item):
                            self.gather.stage(rnd, item, handle, state['ctx'], offset=k * self.item_bytes, nbytes=self.item_bytes)
                    else:
                        def stage(k, handle):
                            return None
                    self.do_step(state, i, stage)
                    done[i] = time.perf_counter()
                    staged[i].set()
            except BaseException as ex:           # never leave the main thread waiting for a dead worker
                errors.append(ex)
                ready.abort()
                for ev in staged:
                    ev.set()
                with cond:
                    cond.notify_all()

        threads = [threading.Thread(target=worker, daemon=True) for _ in range(M)]
        for t in threads:
            t.start()
        try:
            ready.wait()
        except threading.BrokenBarrierError:
            group.any_over_ranks(True)
            raise RuntimeError('a worker thread failed during set-up: %r' % (errors[:1], ))
        if group.any_over_ranks(False):
            raise RuntimeError('set-up failed on another rank')
        ctx = _hip.default_context()
        ctx.synchronize()
        group.barrier()
        t_start = time.perf_counter()
        go.set()
        gather_done = [None] * rounds
        if self.gather is not None:
            for rnd in range(rounds):
                for i in range(rnd * M, min((rnd + 1) * M, total)):
                    staged[i].wait()
                self.gather.flush(rnd)            # collective; a failed rank sends what its ring holds, see any_over_ranks below
                gather_done[rnd] = time.perf_counter()
                with cond:
                    flushed[0] = rnd + 1
                    cond.notify_all()
        for t in threads:
            t.join()
        ctx.synchronize()
        group.barrier()
        t_end = time.perf_counter()
        failed = group.any_over_ranks(bool(errors))
        if errors:
            raise RuntimeError('a worker thread failed: %r' % (errors[0], ))
        if failed:
            raise RuntimeError('a worker thread failed on another rank')
        order = sorted(done)
        t0 = order[warmup - 1] if warmup > 0 else t_start
        t1 = order[warmup + steps - 1]
        if self.gather is not None:               # the gather of the round that holds the last timed step belongs to it
            t1 = max(t1, gather_done[(warmup + steps - 1) // M])
        self.rank_seconds = group.gather_objects(float(t1 - t0))          # (rank 0: every rank's own timed window)
        elapsed = group.max_over_ranks(t1 - t0)
        cold = group.max_over_ranks(t_end - t_start) / total
        return elapsed, cold

    def verify_round(self, state, do_step):
        """one extra, untimed round: every rank runs `do_step` once (item 0 of the round), the round is gathered, and rank 0
        gets every rank's staged maps as host arrays: list over ranks of uint8 buffers of items_per_step * item_bytes"""
        if self.gather is None:
            return None
        rnd = self.gather.rounds_done
        g = self.gather

        def stage(k, handle):
            g.stage(rnd, 0, handle, state['ctx'], offset=k * self.item_bytes, nbytes=self.item_bytes)

        do_step(state, -2, stage)
        out = g.flush(rnd)
        step_bytes = self.item_bytes * self.items_per_step
        if self.group.rank != 0:
            return None
        if out is not None:                        # host plane: [rank][slot][part]
            return [np.concatenate([np.ascontiguousarray(part).view(np.uint8).ravel() for part in items[0]]) for items in out]
        host = np.empty((self.group.world, g.round_bytes), dtype=np.uint8)
        g.ctx.copy(host.ctypes.data, g.recv, host.nbytes, synchronize=True)
        return [host[r, :step_bytes] for r in range(self.group.world)]

    def close(self):
        if self.gather is not None:
            self.gather.close()


def host_link_rate(ctx, image, height, width, sync_group=None, reps=10):
    """GB/s of one step's transfers (the image up, the int32 segmentation down) copied back to back on one stream; with
    `sync_group` every rank starts at a barrier, so the figure is what the link of this rank gives while all ranks copy"""
    import ctypes as C
    from pyimsegm_amd import _hip
    dev = C.c_void_p()
    nbytes_up, nbytes_down = image.nbytes, height * width * 4
    _hip._check(_hip.load_library().imsegm_device_alloc(ctx.device if hasattr(ctx, 'device') else 0, max(nbytes_up, nbytes_down) + 256, C.byref(dev)))
    try:
        down = _hip.pinned_empty((height, width), np.int32)
        for rep in range(reps + 2):
            if rep == 2:
                if sync_group is not None:
                    sync_group.barrier()
                t = time.perf_counter()
            ctx.copy(dev.value, image.ctypes.data, nbytes_up)
            ctx.copy(down.ctypes.data, dev.value, nbytes_down)
        link_s = (time.perf_counter() - t) / reps
    finally:
        _hip.load_library().imsegm_device_free(dev)
    return {'bytes_per_step': nbytes_up + nbytes_down, 'gb_per_s': round((nbytes_up + nbytes_down) / link_s / 1e9, 1),
            'note': 'H2D of the image + D2H of the int32 segmentation, back to back on one stream'}


def host_image(image, pinned):
    if not pinned:
        return image
    from pyimsegm_amd import _hip
    out = _hip.pinned_empty(image.shape, image.dtype)
    out[...] = image
    return out


# -----------------------------------------------------------------------------------------------------------------
# configs 2 / 3 / 4: colour images through segment_color2d_slic_features_model_graphcut
# -----------------------------------------------------------------------------------------------------------------
def fit_inclusive_line(pipe, images, features, sp_size, repeats):
    """SURVEY 8(d) / BASELINE.md section 3: "GMM (host) reported as its own line AND included in end-to-end".  The reference's
    default per-image mode fits the mixture INSIDE the call (`pipe_color2d_slic_features_model_graphcut`,
    /root/reference/imsegm/pipelines.py:46-110, driver :354); `value` of the line times the pre-fitted form
    (`segment_color2d_...`, pipelines.py:160, driver :405).  Here: the fit-inclusive call, one image after the other (the fit is
    scikit-learn on the host -- GaussianMixture(n_init=9) as graph_cuts.py:98-163 configures it -- identical on the CPU side),
    1 warm-up + `repeats` passes over `images`, median per image."""
    fit_seconds = [0.0]
    fit = pipe.estim_class_model

    def timed_fit(*a, **kw):
        t = time.perf_counter()
        try:
            return fit(*a, **kw)
        finally:
            fit_seconds[0] += time.perf_counter() - t
    pipe.estim_class_model = timed_fit
    try:
        totals, fits = [], []
        for rep in range(repeats + 1):
            for image in images:
                np.random.seed(0)
                fit_seconds[0] = 0.0
                t = time.perf_counter()
                pipe.pipe_color2d_slic_features_model_graphcut(image, NB_CLASSES, features, sp_size=sp_size, sp_regul=SP_REGUL,
                                                               pca_coef=None, use_scaler=True, estim_model='GMM', gc_regul=GC_REGUL,
                                                               gc_edge_type=EDGE_TYPE)
                if rep > 0:
                    totals.append(time.perf_counter() - t)
                    fits.append(fit_seconds[0])
    finally:
        pipe.estim_class_model = fit
    total, fit_s = median_min(totals)[0], median_min(fits)[0]
    npx = images[0].shape[0] * images[0].shape[1]
    return {'what': 'pipe_color2d_slic_features_model_graphcut (reference pipelines.py:46-110): SLIC -> descriptors -> '
                    'GaussianMixture fit on the host (scikit-learn, n_init=9, graph_cuts.py:98-163) -> GraphCut, one image at a time, '
                    'host in -> segm AND segm_soft in host numpy; median of %d calls' % len(totals),
            'ms_per_image_including_fit': round(total * 1e3, 3), 'host_model_fit_ms_per_image': round(fit_s * 1e3, 3),
            'ms_per_image_excluding_fit': round((total - fit_s) * 1e3, 3),
            'value_including_fit': round(npx / total / 1e6, 3), 'unit': 'Mpixels/s'}


def bench_color2d(args, group, cfg, quick=False):
    from pyimsegm_amd import _hip
    from pyimsegm_amd import pipelines as pipe
    from pyimsegm_amd.descriptors import FEATURES_SET_COLOR
    from pyimsegm_amd.graph_cuts import estim_class_model
    from pyimsegm_amd.utilities.synthetic import voronoi_image

    world, rank = group.world, group.rank
    golden4 = load_golden('reference_c4.npz') if cfg == 4 else None
    if cfg == 4:
        (height, width), sp_size, per_step = C4_SHAPE, C4_SP_SIZE, C4_PER_STEP
        seeds = [C4_FIRST_SEED + (rank * per_step + i) % C4_NB_IMAGES for i in range(per_step)]
        images = [voronoi_image(height, width, seed=s) for s in seeds]
        features = FEATURES_SET_COLOR
    else:
        height = width = args.size or HEIGHT
        sp_size, per_step = SP_SIZE, 1
        seeds = [1 + rank]
        images = [voronoi_image(height, width, seed=seeds[0])]
        features = FEATURES_SET_COLOR if cfg == 2 else FEATURES_LM
    images = [host_image(im, args.pinned_input) for im in images]
    # (whole multiples of the images in flight: completions come in groups of that size, and a window that cuts a group in two
    # reads too fast -- config 3 with 3 steps and 2 in flight showed 40 ms per image where 8 steps show 58)
    # (config 4 inside `other_configs`: 96 steps -- a window of 24 steps is 27 ms long and read 3.97 as well as 4.75 Gpixel/s on one box)
    steps = args.steps if (args.steps is not None and not quick) else ({3: 9, 4: 96}[cfg] if quick else {2: 100, 3: 12, 4: 96}[cfg])
    warmup = args.warmup if (args.warmup is not None and not quick) else {2: 3, 3: 1, 4: 6}[cfg]
    # (config 3: 80 against 73 Mpixels/s with two; config 4: steps of 8 images in one launch chain each, three of them in flight --
    # twelve when the images go one by one, --batch-images 0)
    # (config 4, round 6: FOUR batches in flight -- 4.34 - 6.08 against 4.24 - 4.76 Gpixel/s with three, alternating on one box, twelve runs)
    inflight = args.inflight if args.inflight > 0 else {2: 4, 3: 3, 4: 12 if args.batch_images == 0 else 4}[cfg]
    if group.distributed and world > 1:
        # worker threads of a rank: no more than the CPUs it has been placed on (the NUMA node of its GPU, distributed.Group)
        from pyimsegm_amd.distributed import worker_threads_per_rank
        inflight = worker_threads_per_rank(world, inflight)
    npx_step = per_step * height * width

    # class model: fitted once, outside the timed region (the reference's group-model flow); config 4 takes the group model
    # of the reference's own run over the 64 images, so that every rank works with the same model and every label map
    # can be checked against that run
    np.random.seed(0)
    ctx = _hip.default_context()
    if golden4 is not None:
        model = model_from_arrays(golden4)
        model_source = 'group model of the reference run over the 64 images (tests/golden/reference_c4.npz)'
    else:
        res0 = pipe._ResidentImage(images[0], features, sp_size, SP_REGUL)
        model = estim_class_model(res0.features, NB_CLASSES, 'GMM', None, True)
        res0.close()
        model_source = 'fitted on the first image of the rank during warm-up'
    on_device = pipe._device_gmm(model) is not None
    classes = getattr(model, 'classes_', None)

    def one_image(image, to_host=True, session=None, stage=None, k=0):
        if to_host and session is None:
            # the library's one-call form of segment_color2d_slic_features_model_graphcut (recycled session); with several
            # ranks the label map is copied into the send ring (device to device) before the session is recycled
            hook = (lambda sess: stage(k, _hip.segm_device_array(sess))) if (stage is not None and group.distributed) else None
            fast = pipe._segment_color2d_one_call(image, model, features, sp_size, SP_REGUL, GC_REGUL, EDGE_TYPE, want_soft=False,
                                                  reuse=True, with_session=hook)
            if fast is not None:
                return fast[0]
        res = pipe._ResidentImage(image, features, sp_size, SP_REGUL, session=session, reuse=session is None,
                                  features_to_host=not on_device)
        try:
            segm, _ = res.segment(None, GC_REGUL, EDGE_TYPE, classes=classes, to_host=to_host, want_soft=False, model=model)
            if stage is not None and group.distributed:
                stage(k, _hip.segm_device_array(res.sess))           # D2D into the send ring before the session is recycled
            return segm
        finally:
            res.close()

    def make_state():
        return {'ctx': _hip.default_context()}

    # config 4: the images of a step go through the library TOGETHER (csrc/batch.hip: every kernel launched once for the
    # batch, image = blockIdx.z) -- what the reference does with a pool map over the images
    batched = cfg == 4 and on_device and args.batch_images != 0 and pipe.BATCH_IMAGES > 0

    # (measured on one GPU, round 4: the ring LOSES -- config 2 5.1 against 6.9 Gpixel/s, config 4 3.75 against 3.9: the runtime's
    # own staging of a pageable source is at least as good as a copy by the worker thread.  That was ONE rank; with N ranks the
    # runtime's staging threads of all ranks share the host's memory system, so with `--input-ring -1` (the default) and more than
    # one rank the choice is MEASURED below -- about a second per form, all ranks at once -- instead of carried over.)
    staging = {'on': args.input_ring == 1 and not args.pinned_input, 'chosen_by': 'flag' if args.input_ring in (0, 1) else 'default (one rank: measured in round 4)'}

    def staged(state, k, image):
        """the image in the thread's own page-locked buffer (allocated once per thread and slot)"""
        if not staging['on'] or state is None:
            return image
        ring = state.setdefault('ring', {})
        buf = ring.get(k)
        if buf is None:
            buf = ring[k] = _hip.pinned_empty(image.shape, image.dtype)
        np.copyto(buf, image)
        return buf

    def do_step(state, index, stage):
        if staging['on']:
            step_images = [staged(state, k, image) for k, image in enumerate(images)]
        else:
            step_images = images
        if batched:
            hook = None
            if stage is not None and group.distributed:
                def hook(batch):
                    for k in range(len(images)):
                        stage(k, batch.segm_device_array(k))         # D2D into the send ring before the batch object is recycled
            got = pipe._segment_color2d_batch_call(step_images, model, features, sp_size, SP_REGUL, GC_REGUL, EDGE_TYPE, with_batch=hook)
            if got is not None:
                return got
        return [one_image(image, stage=stage, k=k) for k, image in enumerate(step_images)]

    rank_links = None
    if group.distributed and world > 1:
        # what the host link of every rank moves with all ranks copying at once (a slow socket or a shared switch shows here)
        link = None
        try:
            link = round(host_link_rate(ctx, images[0], height, width, sync_group=group)['gb_per_s'], 1)
        except Exception:
            pass
        rank_links = group.gather_objects(link)
    if group.distributed and world > 1 and args.input_ring == -1 and not args.pinned_input:
        # which input staging is faster HERE, with every rank of the node moving its images at the same time: this rank's own
        # steady-state rate without a gather, first as the timed run would start (pageable source), then through the ring
        from pyimsegm_amd.distributed import Group as _G
        probe = {}
        for mode in (False, True):
            staging['on'] = mode
            local = SteadyRun(_G(single=True), inflight, make_state, do_step)
            t_cal, _ = local.run(1, 2 * inflight)
            n_probe = max(inflight, int(min(max(2 * inflight, args.probe_seconds / max(t_cal / (2 * inflight), 1e-6)), 4096)) // inflight * inflight)
            group.barrier()
            t_run, _ = local.run(inflight, n_probe)
            local.close()
            probe['ring' if mode else 'pageable'] = round(n_probe * npx_step / t_run / 1e6, 1)
        staging['on'] = probe['ring'] > 1.03 * probe['pageable']
        staging['chosen_by'] = 'probe on this rank with all %d ranks running (Mpixels/s: %r)' % (world, probe)
        staging['probe'] = probe
    runner = SteadyRun(group, inflight, make_state, do_step, gather_item_bytes=height * width * 4, items_per_step=per_step)
    elapsed, cold = runner.run(warmup, steps)
    rank_rings = group.gather_objects({'input_ring': bool(staging['on']), 'chosen_by': staging['chosen_by']}) if (group.distributed and world > 1) else None

    # ---- every gathered label map against the reference's run (N > 1: the first multi-GPU run validates the RCCL gather)
    verdict = {}
    if cfg != 4 and group.distributed:
        # configs 2 / 3 at N > 1: every rank works on its own image (seed 1 + rank), so what rank 0 received is checked
        # against the CRC of the map the SENDING rank holds in its own host memory (host control plane): the device
        # gather is validated end to end by the first multi-GPU run
        own = []

        def step_with_crc(state, index, stage):
            own.append(crc32(one_image(images[0], stage=stage, k=0)))
        maps = runner.verify_round(make_state(), step_with_crc)
        sent = group.gather_objects(own[-1] if own else None)
        if rank == 0:
            got = [crc32(np.ascontiguousarray(m[:height * width * 4]).view(np.int32).reshape(height, width)) for m in maps]
            verdict['gathered_maps_checked'] = world
            verdict['gathered_maps_equal_senders_own'] = bool(got == list(sent))
            if got != list(sent):
                verdict['gathered_maps_different_ranks'] = [r for r in range(world) if got[r] != sent[r]][:16]
    if cfg == 4 and golden4 is not None:
        want = {int(s): int(c) for s, c in zip(golden4['seeds'], golden4['segm_crc'])}
        if group.distributed:
            maps = runner.verify_round(make_state(), do_step)
            if rank == 0:
                bad = []
                for r, buf in enumerate(maps):
                    for k in range(per_step):
                        seed = C4_FIRST_SEED + (r * per_step + k) % C4_NB_IMAGES
                        got = zlib.crc32(np.ascontiguousarray(buf[k * height * width * 4:(k + 1) * height * width * 4]).tobytes())
                        if got != want[seed]:
                            bad.append((r, seed))
                verdict['gathered_maps_equal_reference_run'] = not bad
                verdict['gathered_maps_checked'] = world * per_step
                if bad:
                    verdict['gathered_maps_different'] = bad[:16]
        if rank == 0:
            slic_want = {int(s): int(c) for s, c in zip(golden4['seeds'], golden4['slic_crc'])}
            check_seeds = list(range(C4_FIRST_SEED, C4_FIRST_SEED + C4_NB_IMAGES)) if (world == 1 and not quick) else seeds
            bad_segm, bad_slic = [], []
            for s in check_seeds:
                img = images[seeds.index(s)] if s in seeds else voronoi_image(height, width, seed=s)
                res = pipe._ResidentImage(img, features, sp_size, SP_REGUL, reuse=True, features_to_host=False)
                try:
                    segm, _ = res.segment(None, GC_REGUL, EDGE_TYPE, classes=classes, to_host=True, want_soft=False, model=model)
                    if crc32(res.slic) != slic_want[s]:
                        bad_slic.append(s)
                    if crc32(segm) != want[s]:
                        bad_segm.append(s)
                finally:
                    res.close()
            if batched:
                # ... and the batched path itself: every image of this rank's step, segmentation and superpixel map
                maps = []
                got = pipe._segment_color2d_batch_call(images, model, features, sp_size, SP_REGUL, GC_REGUL, EDGE_TYPE,
                                                       with_batch=lambda b: maps.extend(b.get_labels(k) for k in range(len(images))))
                for k, s_ in enumerate(seeds):
                    if got is None or crc32(got[k]) != want[s_]:
                        bad_segm.append(('batch', s_))
                    if got is None or crc32(maps[k]) != slic_want[s_]:
                        bad_slic.append(('batch', s_))
            verdict['gpu_equals_reference_run'] = not bad_segm and not bad_slic
            verdict['reference_run'] = ('tests/golden/reference_c4.npz (%s): %d images checked (seeds %d..%d), superpixel maps different: %s, '
                                        'segmentations different: %s' % (str(golden4['versions']), len(check_seeds), check_seeds[0],
                                                                          check_seeds[-1], bad_slic or 'none', bad_segm or 'none'))
    runner.close()
    runner_rank_seconds = runner.rank_seconds or []
    placements = group.gather_objects(getattr(group, 'placement', None)) if (group.distributed and world > 1) else None

    # ---- separate figures: device-resident rate, soft D2H, single-image latency (un-timed for `value`)
    extras = {}
    if rank == 0:
        from pyimsegm_amd.superpixels import _open_session
        t = time.perf_counter()
        for _ in range(3):
            one_image(images[0])
        extras['latency_ms'] = round((time.perf_counter() - t) / 3 * 1e3, 3)
        if cfg != 4 and not quick:
            resident = {'sessions': []}

            def make_resident():
                sess, mode = _open_session(images[0])
                st = {'ctx': _hip.default_context(), 'session': (sess, mode)}
                resident['sessions'].append(sess)
                return st

            def resident_step(state, index, stage):
                one_image(images[0], to_host=False, session=state['session'])

            from pyimsegm_amd.distributed import Group as _G
            rr = SteadyRun(_G(single=True), inflight, make_resident, resident_step)     # this process only
            r_elapsed, _ = rr.run(min(warmup, 3), min(steps, 30))
            extras['device_resident'] = {'value': round(min(steps, 30) * npx_step / r_elapsed / 1e6, 3), 'unit': 'Mpixels/s',
                                         'ms_per_step': round(r_elapsed / min(steps, 30) * 1e3, 4),
                                         'note': 'same pipeline with the image already in HBM and `segm` left in HBM'}
            for sess in resident['sessions']:
                sess.close()
            # segm_soft (H x W x C float64, 100 MB at 2048^2) to the host on request
            res = pipe._ResidentImage(images[0], features, sp_size, SP_REGUL, features_to_host=not on_device)
            res.segment(None, GC_REGUL, EDGE_TYPE, classes=classes, to_host=True, want_soft=True, model=model)   # buffers
            t = time.perf_counter()
            res.segment(None, GC_REGUL, EDGE_TYPE, classes=classes, to_host=True, want_soft=False, model=model)
            t_a = time.perf_counter()
            res.segment(None, GC_REGUL, EDGE_TYPE, classes=classes, to_host=True, want_soft=True, model=model)
            t_b = time.perf_counter()
            res.close()
            extras['soft_d2h_ms'] = round(((t_b - t_a) - (t_a - t)) * 1e3, 3)
            # what the host link allows: the bytes of one step (image up, int32 segmentation down) copied back to back on one
            # stream -- on the boxes of this pool the two directions share ~55 GB/s (tools/xfer_concurrent.py: more threads or
            # copying both ways at once moves no more), so this is a ceiling of the host -> host rate, whatever the kernels do

    if rank == 0:
        # what the host link allows: the bytes of one image (pixels up, int32 segmentation down) copied back to back on one
        # stream -- on the boxes of this pool the two directions share ~55 GB/s (tools/xfer_concurrent.py: more threads or
        # copying both ways at once moves no more), so this is a ceiling of the host -> host rate, whatever the kernels do.
        # For every configuration (round 6: config 4 too -- the driver's line of round 5 read 3.6 Gpixel/s where the builder's
        # boxes gave 4.6 - 4.9; the link of the box is the first thing to look at)
        try:
            extras['host_link'] = host_link_rate(ctx, np.asarray(images[0]), height, width)
            extras['host_link']['ceiling_mpixels_per_s'] = round(height * width / (extras['host_link']['bytes_per_step'] / extras['host_link']['gb_per_s'] / 1e9) / 1e6, 1)
            extras['host_link']['fraction_of_ceiling'] = None       # (filled in below, once `value` is known)
        except Exception as ex:
            extras['host_link'] = {'error': repr(ex)}

    # ---- kernel-level figures: un-overlapped pass on one stream with HIP events around every stage
    prof_steps = 3 if cfg == 3 else 5
    ctx.profile_enable(True)
    ctx.profile_reset()
    for _ in range(prof_steps):
        if batched:
            do_step(None, -1, None)                   # (the launch chain of the timed steps: per_step images per launch)
        else:
            one_image(images[0])
    ctx.synchronize()
    stage_ms = {g: ctx.profile_get(g) for g in _hip.PROFILE_GROUPS}
    ctx.profile_enable(False)

    out = None
    if rank == 0:
        value = world * steps * npx_step / elapsed / 1e6
        npx = height * width
        if cfg == 3:
            tex_ms, tex_n = stage_ms['texture']
            flops = 2.0 * 1089 * 76 * 3 * npx                      # 2 * 33^2 taps * 76 kernels * 3 channels per pixel
            achieved = flops / (tex_ms / prof_steps / 1e3) / 1e12 if tex_n else 0.0
            # what the kernels execute since round 4 (flops = 2 per multiply-add, 1 per addition), per output pixel and channel:
            #   the 48 dense kernels are 8 batteries of 3 mirror pairs of point-symmetric kernels (k_conv_battery_quad): per lane and 4
            #   output rows, each of the 16 column pairs costs 72 additions (column sums / differences), 68 x 2 additions (U, V) and
            #   68 x 6 multiply-adds, the centre column 68 x (1 + 3): (16 * 1024 + 476) / 4 = 4215 flops per pixel and battery;
            #   28 kernels are separable (36 rank-1 components: 4 Gaussians, 8 x 2 for the Laplacians, 16 axis-aligned edge / bar
            #   filters) on a 16 x 96 tile: an x pass over 128 rows + a y pass over 96 = (128 + 96) / 96 * 33 = 77 multiply-adds each
            executed = (8 * 4215.0 + 36 * 2.0 * 77) * 3 * npx
            done = executed / (tex_ms / prof_steps / 1e3) / 1e12 if tex_n else 0.0
            roofline = {'bound': 'fp64_valu',
                        'kernel': 'k_conv_battery_quad<3> + k_sep_battery_tall (all 20 batteries of the Leung-Malik bank per image)',
                        'achieved': round(done, 3), 'peak': FP64_VALU_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                        'frac': round(done / FP64_VALU_PEAK_TFLOPS, 5), 'traffic': None,
                        'executed_flops_per_image': executed, 'battery_ms_per_image': round(tex_ms / prof_steps, 3),
                        'note': 'achieved = the flops the kernels EXECUTE over the measured time.  SURVEY 8(d) counts every kernel as a '
                                'dense 33 x 33 sum (2 * 1089 * 76 * 3 flop per pixel); 28 kernels are separable and the other 48 are point '
                                'symmetric mirror pairs, which cuts the work to 24 % of that count -- see survey_flops_*',
                        'survey_flops_per_image': flops, 'survey_flops_effective_tflops': round(achieved, 3)}
        else:
            assign_ms, assign_n = stage_ms['slic_assign']
            sweeps = _hip.assign_sweeps_per_launch() if hasattr(_hip, 'assign_sweeps_per_launch') else 1
            avg_s = assign_ms / max(assign_n, 1) / 1e3
            per_launch = per_step if batched else 1          # images one launch of the kernel works on
            achieved = ASSIGN_BYTES_PER_PX * npx * per_launch * sweeps / avg_s / 1e9 if assign_n else 0.0
            traffic = None
            # HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes (tools/profile_round4.sh): one
            # 2048^2 image per launch, or the eight 647 x 1024 images of a config-4 step in one launch
            traffic_file = 'pmc_traffic_cfg4_batch_r05.json' if (cfg == 4 and batched and per_step == 8) else 'pmc_traffic.json'
            try:
                with open(os.path.join(ROOT, 'profiles', traffic_file)) as fp:
                    pmc = json.load(fp)
                if ((height, width) == (HEIGHT, WIDTH) or traffic_file != 'pmc_traffic.json') and int(pmc.get('sweeps_per_launch', 1)) == sweeps:
                    traffic = pmc['hbm_bytes_per_launch']
            except Exception:
                pass
            kernel = ('k_slic_sweeps (ONE persistent launch for all %d sweeps: per-tile candidate lists, assignment, fused centroid '
                      'accumulation and centroid update)' % sweeps) if sweeps > 1 else \
                'k_slic_assign_dot (assignment + fused centroid accumulation; all sweeps)'
            roofline = {'bound': 'hbm', 'kernel': kernel,
                        'achieved': round(achieved, 2), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                        'frac': round(achieved / HBM_PEAK_GBS, 5), 'traffic': traffic,
                        'traffic_source': 'profiles/%s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes)' % traffic_file,
                        'avg_kernel_us': round(avg_s * 1e6, 3), 'launches': assign_n, 'sweeps_per_launch': sweeps,
                        'images_per_launch': per_launch,
                        'algorithmic_bytes_per_launch': ASSIGN_BYTES_PER_PX * npx * per_launch * sweeps}
        workload = {
            2: 'single %dx%d RGB uint8 per GPU, SLIC(sp_size=46, K=2025, 10 sweeps) + colour mean/std/energy + 3-class '
               'alpha-expansion GC (gc_regul=2.0, edge=model), pre-fitted GMM, host numpy in -> segm int32 in host numpy '
               '(BASELINE configs[1])' % (height, width),
            3: 'single %dx%d RGB uint8 per GPU, SLIC(sp_size=46) + full Leung-Malik bank (76 kernels 33x33, sigma=150 high-pass, '
               'F=180) + 3-class GC, pre-fitted GMM, host in -> host out (BASELINE configs[2])' % (height, width),
            4: 'batch of %d images %dx%d RGB uint8 per GPU per step (BASELINE configs[3]: 64 images over 8 GPUs, seeds 100..163), '
               'SLIC(sp_size=35) + colour mean/std/energy + 3-class GC, pre-fitted GMM, host in -> host out'
               % (per_step, height, width),
        }[cfg]
        out = {
            'metric': METRIC,
            'value': round(value, 3), 'unit': 'Mpixels/s', 'n_gpus': world, 'steps': steps, 'warmup': warmup,
            'ms_per_step': round(elapsed / steps * 1e3, 4), 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f64', 'data': 'synthetic',
            'config': {
                'workload': workload, 'bench_config': cfg, 'images_per_step_per_gpu': per_step,
                'images_per_launch_chain': per_step if batched else 1,
                'images_in_flight_per_gpu': inflight * (per_step if batched else 1) if cfg == 4 else inflight, 'hardware_queues': os.environ.get('GPU_MAX_HW_QUEUES', 'runtime default (4)'),
                'timed_region': 'host numpy image -> H2D -> SLIC -> descriptors -> class model -> graph-cut terms -> '
                                'alpha-expansion -> gathers -> D2H -> segm in host numpy (page-locked result array); model fit outside',
                'input_memory': 'page-locked' if args.pinned_input else ('pageable numpy, copied by the worker thread into its page-locked '
                                                                         'ring inside the timed region' if staging['on'] else 'pageable numpy'),
                'input_ring': staging['chosen_by'],
                'class_model': ('device (scaler + full-covariance GMM)' if on_device else 'host scikit-learn predict_proba') + '; ' + model_source,
                'timing': 'steady state: %d warm-up + %d timed + %d cool-down steps back to back, clock from completion of '
                          'step W to completion of step W+K' % (warmup, steps, inflight),
                'parallelism': 'images sharded over %d GPU(s), %d in flight per GPU (one HIP stream each)%s'
                               % (world, inflight, ', label maps gathered in rank 0 HBM per round of %d steps (%s)'
                                  % (inflight, group.backend) if group.distributed else ''),
            },
            'ms_per_step_incl_fill_drain': round(cold * 1e3, 4),
            'gather_backend': group.backend if group.distributed else None,
            'per_rank': ({'value': [round(steps * npx_step / sec / 1e6, 1) for sec in runner_rank_seconds],
                          'host_link_gb_per_s': rank_links, 'placement': placements, 'input_staging': rank_rings,
                          # what a rank asks of the HOST's memory system at its measured rate: the image read (twice more when it
                          # goes through the ring: the copy's read and write) and the int32 map written, per second
                          'host_memory_traffic_gb_per_s': [
                              round((per_step * (images[0].nbytes * (3 if (rr or {}).get('input_ring') else 1) + height * width * 4))
                                    * steps / sec / 1e9, 4) for sec, rr in zip(runner_rank_seconds, rank_rings or [None] * world)]}
                         if (group.distributed and world > 1) else None),
            'rccl_error': getattr(group, 'rccl_error', None),
            'roofline': roofline,
            'stage_ms_per_step': {g: round(ms / prof_steps, 4) for g, (ms, n) in stage_ms.items()},
            'stage_note': 'stage and roofline figures: separate pass of %d un-overlapped host-to-host %s on one stream'
                          % (prof_steps, 'steps (batches of %d images in one launch chain)' % per_step if batched else 'images'),
        }
        out.update(extras)
        out.update(verdict)
        if isinstance(out.get('host_link'), dict) and out['host_link'].get('ceiling_mpixels_per_s'):
            out['host_link']['fraction_of_ceiling'] = round(value / world / out['host_link']['ceiling_mpixels_per_s'], 3)
        if world == 1:
            try:      # the fit-inclusive form of the same step (its own line item; `value` stays the pre-fitted form)
                n_fit = {2: 3, 3: 1, 4: 1}[cfg] if quick else {2: 5, 3: 2, 4: 1}[cfg]
                inc = fit_inclusive_line(pipe, [np.asarray(im) for im in images], features, sp_size, n_fit)
                out['fit_inclusive'] = inc
                out['host_model_fit_ms_per_step'] = round(inc['host_model_fit_ms_per_image'] * per_step, 3)
                out['ms_per_step_including_fit'] = round(inc['ms_per_image_including_fit'] * per_step, 3)
            except Exception as ex:
                out['fit_inclusive'] = {'error': repr(ex)}
        if world == 1 and not args.no_cpu_baseline and cfg == 2 and not quick:
            try:
                base, segm_cpu, slic_real = cpu_baseline_color2d(np.asarray(images[0]), model, sp_size, SP_REGUL)
                out['cpu_baseline'] = base
                segm_gpu = one_image(images[0])
                out['gpu_equals_cpu_oracle'] = bool(np.array_equal(segm_gpu, segm_cpu))
                out['speedup_vs_cpu_baseline'] = round(value / base['value'], 2)
                if slic_real is not None:
                    res = pipe._ResidentImage(images[0], features, sp_size, SP_REGUL)
                    out['gpu_slic_equals_scikit_image'] = bool(np.array_equal(res.slic, slic_real))
                    res.close()
            except Exception as ex:   # the baseline is a reported extra, never a reason to lose the line
                out['cpu_baseline'] = {'error': repr(ex)}
        if world == 1 and not args.no_cpu_baseline and cfg == 3 and (height, width) == (HEIGHT, WIDTH):
            try:
                out['cpu_baseline'] = cpu_baseline_texture(np.asarray(images[0]), sp_size, SP_REGUL, budget_s=12.0 if quick else 20.0)
                out['speedup_vs_cpu_baseline'] = round(value / out['cpu_baseline']['value'], 1)
            except Exception as ex:
                out['cpu_baseline'] = {'error': repr(ex)}
        if world == 1 and not args.no_cpu_baseline and cfg == 4:
            try:
                arrays = {k: np.array(golden4[k]) for k in ('scaler_mean', 'scaler_scale', 'gmm_weights', 'gmm_means', 'gmm_covariances',
                                                            'gmm_precisions_cholesky')} if golden4 is not None else None
                if arrays is not None:
                    out['cpu_baseline'] = cpu_baseline_batch(arrays, sp_size, SP_REGUL)
                    out['speedup_vs_cpu_baseline'] = round(value / out['cpu_baseline']['value'], 2)
            except Exception as ex:
                out['cpu_baseline'] = {'error': repr(ex)}
        if world == 1 and (height, width) == (HEIGHT, WIDTH) and cfg in (2, 3):
            try:      # the reference's own run on this very image (build container, tests/golden/make_golden_*.py)
                verdict = (compare_config2 if cfg == 2 else compare_config3)(np.asarray(images[0]), pipe)
                if verdict is not None:
                    out.update(verdict)
            except Exception as ex:
                out['reference_run_error'] = repr(ex)
    return out


# -----------------------------------------------------------------------------------------------------------------
# config 5: one gray volume through pipe_gray3d_slic_features_model_graphcut
# -----------------------------------------------------------------------------------------------------------------
def bench_volume(args, group, shape=None, quick=False):
    from pyimsegm_amd import _hip
    from pyimsegm_amd import pipelines as pipe
    from pyimsegm_amd.utilities.synthetic import config5_volume

    shape = tuple(shape) if shape is not None else tuple(int(v) for v in args.volume.split(','))
    steps = args.steps if (args.steps is not None and not quick) else 2
    warmup = args.warmup if (args.warmup is not None and not quick) else 1
    vol = config5_volume(shape, seed=5 + group.rank)                      # float32: SURVEY 8(d) C5
    feats = {'color': ('mean', 'std', 'energy')}
    ctx = _hip.default_context()
    p = C5_PARAMS

    def step():
        np.random.seed(0)
        return pipe.pipe_gray3d_slic_features_model_graphcut(vol, NB_CLASSES, feats, spacing=p['spacing'], sp_size=p['sp_size'],
                                                             sp_regul=p['sp_regul'], gc_regul=p['gc_regul'])

    # the mixture fit is scikit-learn on the host, as in the reference (nine restarts of k-means + EM on K x 3 features): timed
    # apart, it is most of the step at full size
    fit_seconds = [0.0]
    fit = pipe.estim_class_model
    # threads of the host-side fit: the reference does not pin them; with one rank on the box the fit (k-means + EM restarts on
    # K x 3 features, most of the step at full size) may use the host's cores (capped: 3 * 10^5 rows do not feed 256 threads)
    fit_threads = 1
    limiter = None
    if group.world == 1:
        try:
            from threadpoolctl import threadpool_limits
            fit_threads = max(1, min(32, os.cpu_count() or 1))
            limiter = threadpool_limits(limits=fit_threads)
        except Exception:
            fit_threads, limiter = 1, None

    fit_lock = threading.Lock()

    def timed_fit(*a, **kw):
        t = time.perf_counter()
        try:
            return fit(*a, **kw)
        finally:
            with fit_lock:
                fit_seconds[0] += time.perf_counter() - t
    pipe.estim_class_model = timed_fit
    # Volumes in flight (round 6).  A step is device work (upload, supervoxels, labelling, statistics: ~0.35 s at full size), then
    # the mixture fit on the HOST (~0.6 s, the device idle), then terms / cut / gather / download (~0.15 s).  With three volumes in
    # flight -- three worker threads, one HIP stream and one resident session (~75 GB) each, what the 2-D configurations have done
    # since round 2 with four images -- the fits (two at a time, graph_cuts._SideBySideGate) run under the device work of the other
    # volumes: measured 1.10 s per volume alone, 0.77 - 0.81 s with two and 0.52 - 0.55 s with three in flight (one box, s16 / s18 of
    # tools/).  `latency_ms` below is one volume alone.
    inflight = args.inflight if args.inflight > 0 else (3 if int(np.prod(shape)) >= (1 << 28) else 1)
    if inflight > 1:
        try:                                    # (a device without room for two resident volumes: one at a time)
            free_b, total_b = _hip.mem_info()
            while inflight > 1 and free_b < inflight * 85.0 * float(np.prod(shape)):
                inflight -= 1
        except Exception:
            inflight = 1
    # one volume alone first (warm-up + one timed call on this thread): the latency of the call and the share of the fit in it
    step()
    ctx.synchronize()
    latency_ms, latency_fit_ms = float('inf'), 0.0
    segm = None
    for _ in range(2):                          # (the faster of two)
        fit_seconds[0] = 0.0
        t_lat = time.perf_counter()
        segm = step()
        if (time.perf_counter() - t_lat) * 1e3 < latency_ms:
            latency_ms, latency_fit_ms = (time.perf_counter() - t_lat) * 1e3, fit_seconds[0] * 1e3
    classes_found = int(len(np.unique(segm[::4, ::16, ::16])))
    del segm
    # the stage figures: one more un-overlapped volume on the session the latency was measured on, HERE -- behind the run with volumes
    # in flight this thread's session is gone and its 90 GB are allocated again between the stage's events (seconds, once)
    ctx.profile_enable(True)
    ctx.profile_reset()
    step()
    stage_ms = {g: ctx.profile_get(g) for g in _hip.PROFILE_GROUPS}
    ctx.profile_enable(False)
    if inflight > 1:
        ctx.close_idle_sessions()               # (the worker threads bring their own resident volumes)
    def do_step(state, index, stage):
        step()                                  # (the class map goes back to the pool of page-locked result arrays at once)
    if inflight > 1 and (args.steps is None or quick):
        steps, warmup = max(steps, 2 * inflight), max(warmup, inflight)
    runner = SteadyRun(group, inflight, lambda: {'ctx': _hip.default_context()}, do_step)
    fit_seconds[0] = 0.0
    elapsed, cold = runner.run(warmup, steps)
    runner.close()
    if inflight > 1:
        _hip.reap_contexts()                    # (the resident volumes of the worker threads that have just ended)
    # (the clock of the fit: its share of the timed steps only -- SteadyRun runs first-use, warm-up, timed and cool-down steps)
    fit_seconds[0] = fit_seconds[0] * steps / (warmup + steps + 2 * inflight)
    fit_ms = fit_seconds[0] / steps * 1e3
    pipe.estim_class_model = fit
    if limiter is not None:
        try:
            limiter.restore_original_limits()
            from threadpoolctl import threadpool_limits
            threadpool_limits(limits=1)           # (what main() set for the image configurations)
        except Exception:
            pass
    if group.rank != 0:
        return None
    nvox = int(np.prod(shape))
    assign_ms, assign_n = stage_ms['slic_assign']
    slic_ms = stage_ms['slic'][0]
    if assign_n:           # the assignment kernel itself (HIP events on its dispatches), 8 B/voxel/sweep
        avg_s = assign_ms / assign_n / 1e3
        roofline = {'bound': 'hbm', 'kernel': 'k_vol_assign_f32 (3-D assignment, all sweeps)',
                    'achieved': round(VOL_ASSIGN_BYTES_PER_VOXEL * nvox / avg_s / 1e9, 2), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                    'frac': round(VOL_ASSIGN_BYTES_PER_VOXEL * nvox / avg_s / 1e9 / HBM_PEAK_GBS, 5), 'traffic': None,
                    'avg_kernel_us': round(avg_s * 1e6, 1), 'launches': assign_n,
                    'algorithmic_bytes_per_launch': VOL_ASSIGN_BYTES_PER_VOXEL * nvox}
        try:
            # HBM bytes and instruction counts of the kernel from the committed rocprofv3 PMC passes (tools/c5_pmc.sh, a quarter
            # volume with the same bricks and windows: per voxel, scaled to this volume).  HBM is not what bounds this kernel:
            # ~60 search windows cover a voxel, a wave of 256 voxels issues ~1 350 vector + ~1 180 scalar instructions -- the floor
            # of the vector unit (4 cycles per wave64 instruction at 2.4 GHz, 1 024 SIMDs) is quoted beside the HBM one.
            with open(os.path.join(ROOT, 'profiles', 'pmc_traffic_cfg5_r06.json')) as fp:
                pmc = json.load(fp)
            roofline['traffic'] = int(pmc['hbm_bytes_per_voxel'] * nvox)
            roofline['traffic_source'] = 'profiles/pmc_traffic_cfg5_r06.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, per voxel x voxels)'
            issue_s = (nvox / 256.0 / 1024.0) * pmc['valu_per_wave'] * 4.0 / 2.4e9
            roofline['valu_issue_floor'] = {'valu_per_wave': round(pmc['valu_per_wave'], 1), 'salu_per_wave': round(pmc['salu_per_wave'], 1),
                                            'floor_us': round(issue_s * 1e6, 1), 'frac_of_floor': round(issue_s / avg_s, 3),
                                            'note': 'waves per SIMD x vector instructions per wave x 4 cycles / 2.4 GHz: what the vector unit '
                                                    'alone needs for the instructions this kernel issues'}
        except Exception:
            pass
    else:
        achieved = 10 * VOL_ASSIGN_BYTES_PER_VOXEL * nvox / (slic_ms / 1e3) / 1e9 if slic_ms else 0.0
        roofline = {'bound': 'hbm', 'kernel': 'whole 3-D SLIC stage (pre-processing, 10 x [scatter, k_vol_assign_f32, k_vol_update_f32], connectivity)',
                    'achieved': round(achieved, 2), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': round(achieved / HBM_PEAK_GBS, 5),
                    'traffic': None, 'algorithmic_bytes': 10 * VOL_ASSIGN_BYTES_PER_VOXEL * nvox, 'stage_ms': round(slic_ms, 2)}
    out = {
        'metric': METRIC,
        'value': round(group.world * steps * nvox / elapsed / 1e6, 3), 'unit': 'Mpixels/s', 'n_gpus': group.world, 'steps': steps,
        'warmup': warmup, 'ms_per_step': round(elapsed / steps * 1e3, 3), 'higher_is_better': True, 'scaling': 'weak',
        'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
        'config': {'workload': 'single %dx%dx%d float32 gray volume per GPU through pipe_gray3d_slic_features_model_graphcut '
                               '(sp_size=15, spacing (1,1,1), gray mean/std/energy, 3-class GMM fitted inside the step on the host, '
                               'gc_regul=0.1), host in -> host out (BASELINE configs[4]); unit = Mvoxels/s' % shape,
                   'volumes_in_flight_per_gpu': inflight,
                   'timing': 'steady state: %d warm-up + %d timed + %d cool-down volumes back to back on %d worker thread(s), clock from the '
                             'completion of step W to the completion of step W+K; latency_ms = one volume alone' % (warmup, steps, inflight, inflight),
                   'bench_config': 5, 'classes_found': classes_found},
        'roofline': roofline,
        'stage_ms_per_step': {g: round(ms, 3) for g, (ms, n) in stage_ms.items()},
        # (one volume alone: the whole call, its host-side fit, and the rest -- with volumes in flight the fit of one runs under the
        # device work of the next, so `ms_per_step` is NOT their sum)
        'host_model_fit_ms_per_step': round(latency_fit_ms, 1), 'ms_per_step_excluding_fit': round(latency_ms - latency_fit_ms, 1),
        'host_model_fit_ms_in_flight': round(fit_ms, 1),      # (per step of the timed run, waiting for the other volume's fit included)
        'volumes_in_flight': inflight, 'latency_ms': round(latency_ms, 1), 'latency_host_model_fit_ms': round(latency_fit_ms, 1),
        'ms_per_step_incl_fill_drain': round(cold * 1e3, 1),
        'host_model_fit_threads': fit_threads,
        'host_model_fit': 'scikit-learn GaussianMixture(full, n_init=9) as graph_cuts.py:73-163 configures it; the restarts of its EM loop '
                          'run side by side (graph_cuts.fit_mixture_restarts: same random stream, same calls, parameters bit for bit '
                          'those of mixture.fit -- tests/test_class_models.py), %d worker threads' % __import__('pyimsegm_amd.graph_cuts', fromlist=['x'])._fit_workers(),
    }
    if group.world == 1:
        try:
            verdict = compare_config5(shape, pipe, vol=vol if group.rank == 0 else None)
            if verdict is not None:
                out.update(verdict)
        except Exception as ex:
            out['reference_run_error'] = repr(ex)
        if not args.no_cpu_baseline:
            try:
                # a 64 x 256 x 256 crop (about ten seconds of one core), the fit included; the reduced
                # volume of `other_configs` is smaller than that crop and is timed whole
                out['cpu_baseline'] = cpu_baseline_volume(shape, crop=(64, 256, 256) if int(np.prod(shape)) > 64 * 256 * 256 else shape)
                out['speedup_vs_cpu_baseline'] = round(out['value'] / out['cpu_baseline']['value'], 2)
            except Exception as ex:
                out['cpu_baseline'] = {'error': repr(ex)}
    del vol
    ctx.close_idle_sessions()                  # (a volume session is ~70 bytes per voxel: nothing of it stays behind the bench)
    return out


# -----------------------------------------------------------------------------------------------------------------
# --gpus N without a launcher: this script starts its own ranks
# -----------------------------------------------------------------------------------------------------------------
def spawn_ranks(n, argv):
    """N processes of this script, one per GPU, with the environment a distributed launcher would set (RANK, LOCAL_RANK,
    WORLD_SIZE, MASTER_ADDR 127.0.0.1, a free MASTER_PORT); rank 0 writes to this process's stdout.  Returns the exit code."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
                   TORCHELASTIC_RUN_ID='bench%d' % os.getpid(), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(sys.argv[0])] + list(argv), env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    code = 0
    for proc in procs:
        code = max(code, abs(proc.wait()))
    return code


def main():
    args = parse_args()
    launched = 'RANK' in os.environ or 'WORLD_SIZE' in os.environ
    if args.gpus is not None and args.gpus > 1 and not launched:
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:]))
    world_env = int(os.environ.get('WORLD_SIZE', '1') or 1)
    if world_env > 1 or (args.config or 2) != 5:
        # host-side BLAS (model fit, non-GMM models): one thread each when N ranks x M worker threads share the node.  Not for
        # the volume configuration on one GPU: its scikit-learn fit is most of the step and the reference does not pin threads
        # (bench_volume lifts the limit for itself when it runs behind config 2 in the default invocation)
        try:
            from threadpoolctl import threadpool_limits
            threadpool_limits(limits=1)
        except Exception:
            pass
    from pyimsegm_amd import _hip
    _hip.init(hardware_queues=8)        # explicit (imsegm_init): four images in flight + the default stream want more than 4 queues
    from pyimsegm_amd.distributed import Group
    group = Group()
    if args.gpus is not None and args.gpus != group.world:
        raise SystemExit('--gpus %d but the launcher started %d rank(s)' % (args.gpus, group.world))
    try:
        cfg = args.config or 2
        out = bench_volume(args, group) if cfg == 5 else bench_color2d(args, group, cfg)
        if args.config is None and not args.no_other_configs and args.size is None:
            # the other BASELINE configurations, driver-timed in the same invocation (a few seconds each): config 4 at any N (the
            # configuration that shards over the GPUs), configs 3 and the reduced 5 on one GPU
            others = {}
            for name, fn in (('4', lambda: bench_color2d(args, group, 4, quick=True)),
                             ('3', lambda: bench_color2d(args, group, 3, quick=True) if group.world == 1 else None),
                             ('5_reduced', lambda: bench_volume(args, group, shape=C5_REDUCED, quick=True) if group.world == 1 else None),
                             # BASELINE configs[4] at its own size (64 x 4096 x 4096: about three minutes, most of it the host
                             # generating the 10^9 voxels): the supervoxel map against the real scikit-image's on the full volume
                             ('5', lambda: bench_volume(args, group, shape=C5_FULL, quick=True) if (group.world == 1 and not args.no_full_volume) else None)):
                try:
                    t0 = time.perf_counter()
                    res = fn()
                    if res is not None:
                        res['wall_s'] = round(time.perf_counter() - t0, 2)
                        for key in ('metric', 'higher_is_better', 'vs_baseline', 'data', 'scaling'):
                            res.pop(key, None)
                        others[name] = res
                except Exception as ex:
                    if group.world > 1:
                        raise
                    others[name] = {'error': repr(ex)}
            if group.rank == 0:
                out['other_configs'] = others
        if group.rank == 0:
            assert out['n_gpus'] == (args.gpus or group.world), (out['n_gpus'], args.gpus, group.world)
        # the communicator goes first, and whatever native libraries left in the C stdio buffer (RCCL prints a version banner
        # to stdout) is flushed, so that the JSON line is the LAST line this process writes
        group.close()
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        if group.rank == 0:
            print(json.dumps(out), flush=True)
    finally:
        group.close()


if __name__ == '__main__':
    main()
