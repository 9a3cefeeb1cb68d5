"""The C oracle's SLIC sweeps and connectivity enforcement against a literal, loop-for-loop pure-Python
restatement of scikit-image 0.18's `_slic_cython` / `_enforce_label_connectivity_cython` (the third-party code
behind /root/reference/imsegm/superpixels.py:61-63,104-106).  This pins the oracle's control flow -- windows,
visiting order, tie-breaking, BFS order, size caps -- on inputs small enough for Python loops; the outputs of the real
scikit-image 0.18.3 are pinned separately in tests/test_golden_skimage.py.

The literal version adds the colour sums sequentially in fp64 exactly like `_slic.pyx`; the oracle adds them
as exact fixed-point sums.  The two agree to ~1e-15, which only matters for a pixel whose two best centroids
are closer than that; the inputs below are random floats, where this does not occur."""
import numpy as np
import pytest


def literal_slic(pre, segments, isteps, step, spacing, max_iter, slic_zero=False):
    """pre: (C, D, H, W) pre-processed image; segments: (K, 3 + C) initial centroids (modified in place)"""
    nch, D, H, W = pre.shape
    K = segments.shape[0]
    step_z, step_y, step_x = isteps
    nearest = np.full((D, H, W), -1, dtype=np.int64)
    sz, sy, sx = spacing
    spatial_weight = 1.0 / (step * step)
    max_dist_color = np.ones(K)
    for _ in range(max_iter):
        distance = np.full((D, H, W), np.finfo(np.float64).max)
        for k in range(K):
            cz, cy, cx = segments[k, :3]
            if np.isnan(cz):
                continue
            z_min, z_max = int(max(cz - 2 * step_z, 0)), int(min(cz + 2 * step_z + 1, D))
            y_min, y_max = int(max(cy - 2 * step_y, 0)), int(min(cy + 2 * step_y + 1, H))
            x_min, x_max = int(max(cx - 2 * step_x, 0)), int(min(cx + 2 * step_x + 1, W))
            for z in range(z_min, z_max):
                dz = (sz * (cz - z))**2
                for y in range(y_min, y_max):
                    dy = (sy * (cy - y))**2
                    for x in range(x_min, x_max):
                        dist_center = (dz + dy + (sx * (cx - x))**2) * spatial_weight
                        dist_color = 0.0
                        for c in range(nch):
                            dist_color += (pre[c, z, y, x] - segments[k, 3 + c])**2
                        if slic_zero:
                            dist_center += dist_color / max_dist_color[k]
                        else:
                            dist_center += dist_color
                        if distance[z, y, x] > dist_center:
                            nearest[z, y, x] = k
                            distance[z, y, x] = dist_center
        count = np.zeros(K, dtype=np.int64)
        segments[:, :] = 0
        for z in range(D):
            for y in range(H):
                for x in range(W):
                    k = nearest[z, y, x]
                    if k < 0:
                        continue
                    count[k] += 1
                    segments[k, 0] += z
                    segments[k, 1] += y
                    segments[k, 2] += x
                    for c in range(nch):
                        segments[k, 3 + c] += pre[c, z, y, x]
        with np.errstate(invalid='ignore', divide='ignore'):
            segments /= count[:, None].astype(np.float64)          # 0/0 -> nan: the centroid is dead
        if slic_zero:
            for z in range(D):
                for y in range(H):
                    for x in range(W):
                        k = nearest[z, y, x]
                        if k < 0:
                            continue
                        dist_color = 0.0
                        for c in range(nch):
                            dist_color += (pre[c, z, y, x] - segments[k, 3 + c])**2
                        if max_dist_color[k] < dist_color:
                            max_dist_color[k] = dist_color
    return nearest


def literal_connectivity(segments, min_size, max_size, start_label=0):
    D, H, W = segments.shape
    dd = [(0, 0, 1), (0, 0, -1), (0, 1, 0), (0, -1, 0), (1, 0, 0), (-1, 0, 0)]       # x+1, x-1, y+1, y-1, z+1, z-1
    mask_label = start_label - 1
    out = np.full((D, H, W), mask_label, dtype=np.int64)
    current = start_label
    for z in range(D):
        for y in range(H):
            for x in range(W):
                if out[z, y, x] >= start_label:
                    continue
                adjacent = 0
                label = segments[z, y, x]
                out[z, y, x] = current
                coords = [(z, y, x)]
                visited = 0
                while visited < len(coords) < max_size:
                    pz, py, px = coords[visited]
                    for dz, dy, dx in dd:
                        zz, yy, xx = pz + dz, py + dy, px + dx
                        if 0 <= zz < D and 0 <= yy < H and 0 <= xx < W:
                            if segments[zz, yy, xx] == label and out[zz, yy, xx] == mask_label:
                                out[zz, yy, xx] = current
                                coords.append((zz, yy, xx))
                                if len(coords) >= max_size:
                                    break
                            elif out[zz, yy, xx] >= start_label and out[zz, yy, xx] != current:
                                adjacent = out[zz, yy, xx]
                    visited += 1
                if len(coords) < min_size:
                    for c in coords:
                        out[c] = adjacent
                else:
                    current += 1
    return out


CASES = [
    ('colour2d', (1, 26, 34), 3, 30, 4.0, (1., 1., 1.)),
    ('colour2d_loose', (1, 31, 23), 3, 12, 0.7, (1., 1., 1.)),
    ('gray3d_aniso', (5, 14, 17), 1, 18, 2.0, (3., 1., 1.)),
    ('colour2d_slico', (1, 26, 34), 3, 30, 4.0, (1., 1., 1.)),
    ('colour2d_slico_loose', (1, 29, 31), 3, 14, 0.5, (1., 1., 1.)),
]


@pytest.mark.parametrize('name,shape,nch,n_segments,compactness,spacing', CASES, ids=[c[0] for c in CASES])
def test_oracle_sweeps_and_connectivity_match_literal_restatement(oracle, name, shape, nch, n_segments, compactness,
                                                                  spacing):
    rng = np.random.default_rng(7)
    D, H, W = shape
    slico = 'slico' in name
    if nch == 3:
        image = rng.random((H, W, 3))
        labels, info = oracle.slic(image, n_segments, compactness, sigma=1., return_internals=True, slic_zero=slico)
    else:
        image = rng.random((D, H, W))
        labels, info = oracle.slic(image, n_segments, compactness, sigma=1., spacing=spacing, multichannel=False,
                                   return_internals=True)
    pre = np.asarray(info['pre']).reshape(nch, D, H, W)
    cent, _ = oracle.grid_centroids((D, H, W), n_segments)
    K = cent.shape[0]
    assert K == info['K']
    segments = np.zeros((K, 3 + nch))
    segments[:, :3] = cent
    nearest = literal_slic(pre, segments, info['steps'], float(np.float32(info['step'])), spacing, 10, slic_zero=slico)
    assert np.array_equal(nearest, np.asarray(info['nearest']).reshape(D, H, W)), \
        '%d voxels differ' % np.count_nonzero(nearest != np.asarray(info['nearest']).reshape(D, H, W))
    segment_size = D * H * W / K
    final = literal_connectivity(nearest, int(0.5 * segment_size), int(3 * segment_size))
    assert np.array_equal(final, np.asarray(labels).reshape(D, H, W))
