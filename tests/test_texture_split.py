"""Host side of the separable Leung-Malik kernels (pyimsegm_amd._hip.Image2D._split_battery): which kernels of the bank of
/root/reference/imsegm/descriptors.py:903-948 leave the dense 33 x 33 sums, and that their factors reproduce them."""
import numpy as np


def test_the_separable_kernels_of_the_bank_and_their_factors():
    from pyimsegm_amd import _hip, descriptors as D
    filters, names = D.create_filter_bank_lm_2d()
    assert sum(len(b) for b in filters) == 76
    taken = 0
    for battery, name in zip(filters, names):
        weights, dense, taps, groups, rank, radius, parity = _hip.Image2D._split_battery(battery, mirror=False)
        assert radius == 16 and dense in (0, 6) and taps.shape == (groups, rank, 2, 33)
        kind = name.split('-')[1]
        assert parity == {'edge': -1, 'bar': 1}.get(kind, 0), name        # (no dense kernels left: nothing to be symmetric)
        # the Gaussian: rank 1; both Laplacians of a Gaussian: rank 2; edge / bar: the orientations 0 and 90 degrees, rank 1 each
        assert (groups, rank, dense) == {'Gauss': (1, 1, 0), 'GaussLap': (1, 2, 0), 'GaussLap2': (1, 2, 0), 'edge': (2, 1, 6),
                                        'bar': (2, 1, 6)}[kind], name
        taken += groups
        flipped = np.asarray(battery)[:, ::-1, ::-1]
        which = [0, 4] if len(battery) == 8 else [0]
        for g, k in enumerate(which):
            rebuilt = sum(np.outer(taps[g, i, 1], taps[g, i, 0]) for i in range(rank))          # sum_i y_i x_i^T
            assert np.max(np.abs(rebuilt - flipped[k])) <= 1e-15 * max(1.0, np.abs(flipped[k]).max() * 1e2), name
        if dense:
            rest = [k for k in range(len(battery)) if k not in which]
            assert weights.shape == (33, 33, dense)
            for j, k in enumerate(rest):
                assert np.array_equal(weights[:, :, j], flipped[k].T)                           # layout [kx][ky][kernel]
    assert taken == 28
    # separable=False: everything dense, padded to 1 / 2 / 4 / 6 / 8 kernels by repeating the last one
    weights, dense, taps, groups, rank, _, parity = _hip.Image2D._split_battery(filters[0], separable=False, symmetric=False)
    assert (dense, groups, rank, parity) == (8, 0, 0, 0) and taps.size == 0
    weights, dense, _, groups, _, _, _ = _hip.Image2D._split_battery(np.asarray(filters[0])[1:4], separable=False)
    assert dense == 4 and np.array_equal(weights[:, :, 3], weights[:, :, 2])


def test_mirror_pairs_of_the_bank_and_the_quad_table():
    """the orientations theta and pi - theta of every edge / bar battery are mirror images (Image2D._mirror_pairs), and the table
    the device sums over quads of pixels (Image2D._quad_table, csrc/texture.hip k_conv_battery_quad) gives the responses of
    scipy.ndimage.convolve -- the sums restated in numpy"""
    from scipy import ndimage
    from pyimsegm_amd import _hip, descriptors as D
    rng = np.random.default_rng(3)
    img = rng.random((60, 71))
    r = 16
    for bank in (D.create_filter_bank_lm_2d(), D.create_filter_bank_lm_2d(sigmas=D.SHORT_FILTERS_SIGMAS, nb_orient=4)):
        for battery, name in zip(*bank):
            weights, dense, taps, groups, rank, radius, parity = _hip.Image2D._split_battery(battery)
            kind = name.split('-')[1]
            assert parity == {'edge': -2, 'bar': 2}.get(kind, 0), name
            if not parity:
                continue
            pairs = dense // 2
            assert dense == len(battery) - 2 and weights.size == 33 * 33 * dense
            table = weights[:(r + 1) * (r + 1) * 2 * pairs].reshape(r + 1, r + 1, 2 * pairs)
            signs = weights[table.size:table.size + pairs]
            assert set(np.abs(signs)) == {1.0} and not weights[table.size + pairs:].any()
            s = 1.0 if parity > 0 else -1.0
            padded = np.pad(img, r, mode='symmetric')               # (scipy's 'reflect')
            h, w = img.shape
            acc_s, acc_d = np.zeros((pairs, h, w)), np.zeros((pairs, h, w))
            for x in range(r + 1):
                a, b = padded[:, r - x:r - x + w], padded[:, r + x:r + x + w]
                total, diff = (a if x == 0 else a + b), b - a
                for t in range(r + 1):
                    u = total[t:t + h] + s * total[2 * r - t:2 * r - t + h]
                    v = diff[t:t + h] - s * diff[2 * r - t:2 * r - t + h]
                    for k in range(pairs):
                        acc_s[k] += table[x, t, k] * u
                        acc_d[k] += table[x, t, pairs + k] * v
            got = np.max([acc_s + acc_d, signs[:, None, None] * (acc_s - acc_d)], axis=(0, 1))
            rest = [k for k in range(len(battery)) if k not in (0, len(battery) // 2)]
            ref = np.max([ndimage.convolve(img, battery[k]) for k in rest], axis=0)
            assert np.max(np.abs(got - ref)) <= 1e-13 * max(1.0, np.abs(ref).max()), name
    # kernels that do not pair up keep the point-symmetric form
    filters, _ = D.create_filter_bank_lm_2d()
    assert _hip.Image2D._split_battery(np.asarray(filters[0])[[1, 2, 3, 5]])[6] == -1


def test_descriptor_groups_that_stay_on_the_device():
    """which feature sets pipelines._ResidentImage keeps in one resident table (descriptors.resident_feature_groups): 'color' in RGB
    and 'tLM*' with mean / std / energy, in the column order of compute_selected_features_color2d
    (/root/reference/imsegm/descriptors.py:1207-1270: the colour groups first, then the Leung-Malik ones)"""
    from pyimsegm_amd.descriptors import resident_feature_groups
    assert resident_feature_groups({'color': ('mean', 'median')}) is None
    assert resident_feature_groups({'color_hsv': ('mean', )}) is None
    assert resident_feature_groups({'tLM': ('mean', ), 'unknown': ('mean', )}) is None
    assert resident_feature_groups({'tLM': ()}) is None and resident_feature_groups({}) is None
    groups = resident_feature_groups({'tLM_short': ('energy', 'mean'), 'tLM': ('mean', 'std', 'energy'), 'color': ('std', )})
    assert [(g[0], sorted(g[1]), g[3]) for g in groups] == [('color', ['std'], 3), ('tLM', ['energy', 'mean'], 90),
                                                           ('tLM', ['energy', 'mean', 'std'], 180)]
    assert len(groups[1][2]) == 15 and len(groups[2][2]) == 20 and groups[0][2] is None
