"""Seeded synthetic inputs for the parity tests and ``bench.py``.
There is no network for datasets, so every benchmark image is generated here."""
import numpy as np


def disc_image(size=256, radius=80, seed=0):
    """config C1: background U[20, 60) with one filled disc U[160, 220), uint8 RGB"""
    rng = np.random.default_rng(seed)
    img = rng.integers(20, 60, (size, size, 3))
    yy, xx = np.mgrid[:size, :size]
    mask = (yy - size // 2)**2 + (xx - size // 2)**2 <= radius**2
    fg = rng.integers(160, 220, (size, size, 3))
    img[mask] = fg[mask]
    return img.astype(np.uint8)


def voronoi_image(height=2048, width=2048, nb_seeds=24, seed=1, noise=12., return_classes=False):
    """configs C2-C4: 3-class piecewise-constant image (Voronoi cells of ``nb_seeds`` points,
    class = cell id mod 3) with class colours + N(0, noise) clipped to uint8; ``return_classes``: also the class map
    (the annotation of the supervised path)"""
    rng = np.random.default_rng(seed)
    pts = np.stack([rng.uniform(0, height, nb_seeds), rng.uniform(0, width, nb_seeds)], axis=1)
    yy, xx = np.mgrid[:height, :width]
    best = np.full((height, width), np.inf)
    cell = np.zeros((height, width), dtype=np.int32)
    for i, (py, px) in enumerate(pts):
        d = (yy - py)**2 + (xx - px)**2
        upd = d < best
        best[upd] = d[upd]
        cell[upd] = i
    colours = np.array([(60, 60, 180), (200, 180, 40), (40, 170, 90)], dtype=np.float64)
    img = colours[cell % 3] + rng.normal(0, noise, (height, width, 3))
    img = np.clip(np.round(img), 0, 255).astype(np.uint8)
    return (img, (cell % 3).astype(np.int64)) if return_classes else img


def ellipsoid_volume(shape=(16, 64, 64), seed=5, noise=0.05):
    """config C5 (small version): three nested ellipsoids + N(0, noise), float32 gray volume"""
    rng = np.random.default_rng(seed)
    zz, yy, xx = np.meshgrid(*[np.linspace(-1, 1, s) for s in shape], indexing='ij')
    r = np.sqrt((zz / 0.9)**2 + (yy / 0.8)**2 + (xx / 0.7)**2)
    vol = np.zeros(shape)
    for thr, val in ((0.9, 0.3), (0.6, 0.6), (0.3, 0.9)):
        vol[r < thr] = val
    vol += rng.normal(0, noise, shape)
    return vol.astype(np.float32)


def config5_volume(shape=(64, 4096, 4096), seed=5, noise=0.05):
    """config C5 at any size, float32: `ellipsoid_volume(shape, seed=5)` plus a second float32 noise field drawn from
    `default_rng(seed)` -- the volume `bench.py --config 5` has always used -- generated slab by slab along z so that the
    10^9-voxel volume needs no float64 temporaries of its size (the draws of a Generator are sequential in C order, so
    the slabs reproduce the one-shot arrays bit for bit; tests/test_golden_configs.py checks that on a small shape)"""
    depth, height, width = shape
    rng_a = np.random.default_rng(5)              # ellipsoid_volume's own generator (its default seed)
    rng_b = np.random.default_rng(seed)
    zs = np.linspace(-1, 1, depth)
    yy = (np.linspace(-1, 1, height) / 0.8)**2
    xx = (np.linspace(-1, 1, width) / 0.7)**2
    out = np.empty(shape, dtype=np.float32)
    for z in range(depth):
        r = np.sqrt(((zs[z] / 0.9)**2 + yy[:, None]) + xx[None, :])     # the summation order of ellipsoid_volume
        vol = np.zeros((height, width))
        for thr, val in ((0.9, 0.3), (0.6, 0.6), (0.3, 0.9)):
            vol[r < thr] = val
        vol += rng_a.normal(0, noise, (height, width))
        out[z] = vol.astype(np.float32)
        out[z] += (0.05 * rng_b.standard_normal((height, width), dtype=np.float32))
    return out
