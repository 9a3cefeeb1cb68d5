/*
 * imsegm_oracle.c -- CPU ORACLE (TEST INFRASTRUCTURE ONLY, never shipped / never the measured product).
 *
 * Plain-C restatement of the reference's SLIC -> per-superpixel descriptors -> alpha-expansion
 * GraphCut hot path.  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may load this library.
 *
 * PARITY STATUS (also stated in DESIGN.md).  Besides the per-stage pins below, the whole chain is checked against a run
 * of the reference itself (tests/golden/make_golden_reference.py, tests/test_golden_reference.py):
 *   - descriptors (orc_color2d_*, orc_gray3d_*): restate /root/reference/imsegm/features_cython.pyx:59-219
 *     line by line; pinned against the reference doctest vectors (descriptors.py:218-283, 470-537)
 *     and against the reference .pyx itself compiled into oracle/_ref (bit-exact).
 *   - graph / centres: restate /root/reference/imsegm/superpixels.py:115-242; pinned against the
 *     doctests superpixels.py:163-168,186-193,211-215.
 *   - alpha expansion: restates gco-wrapper >= 3.0.8 (pyGCO `cut_general_graph`, GCO-v3
 *     GCoptimization::expansion(-1) + Kolmogorov energy.h reparametrisation); the third-party
 *     sources are NOT in /root/reference.  Pinned by the reference doctests
 *     graph_cuts.py:700-703 and region_growing.py:72-75.  Beyond those: PARITY UNPINNED.
 *   - SLIC: restates scikit-image 0.18.x `skimage.segmentation.slic` (slic_superpixels.py,
 *     _slic.pyx, util/_regular_grid.py, color/colorconv.py rgb2lab), `measure.label` and the
 *     regionprops centroids -- third-party, absent from /root/reference.  The reference holds no golden
 *     label maps (superpixels.py:32-40 asserts shapes only), but the build container carries a conda
 *     Python 3.9 with scikit-image 0.18.3: tests/golden/skimage.npz holds its outputs for the
 *     reference's call shapes (tests/golden/make_golden_skimage.py) and this oracle reproduces every
 *     label map BIT FOR BIT (tests/test_golden_skimage.py: colour / gray / float 2-D incl. SLICO and the
 *     raw assignment, uint8 / uint16 / float64 volumes incl. anisotropic spacing, measure.label,
 *     centroids)  =>  PINNED.  One exception: a float32 volume runs in float32 inside scikit-image 0.18
 *     (sequential float32 centroid sums); here it is widened to float64, and ~30 % of the voxels of
 *     the test volume end up in a different (equally valid) supervoxel -- documented deviation.
 *     The Gaussian blur step is also pinned bit-exactly against scipy.ndimage.gaussian_filter
 *     (scipy is installed) in tests/test_oracle_slic.py.
 *     Two conscious, documented deviations from skimage's floating-point arithmetic, both below the
 *     resolution of the label maps in every pinned case (needed so
 *     that a massively parallel device implementation can be bit-identical to this oracle):
 *       (1) x^2.4 and cbrt are evaluated with the deterministic division-free Newton routines
 *           below (only + - * and exact bit operations) instead of libm pow/cbrt (<= 6 ulp);
 *       (2) the centroid colour sums are order-independent exact sums of the values in 2^-f fixed
 *           point (f >= 44 for |value| < 4, see orc_fix_bits) instead of the raster-order running
 *           fp64 sum of _slic.pyx: per-pixel truncation < 2^-44 of the value range, the size of
 *           the rounding noise of that running sum itself.
 *     The control flow of the sweeps and of the connectivity pass is checked against a literal
 *     pure-Python restatement of _slic.pyx in tests/test_oracle_slic.py.
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off, no -ffast-math).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>

#define ORC_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------------------------------ */
/* deterministic elementary functions                                                          */
/* ------------------------------------------------------------------------------------------ */

static inline uint64_t d2b(double x) { uint64_t b; memcpy(&b, &x, 8); return b; }
static inline double b2d(uint64_t b) { double x; memcpy(&x, &b, 8); return x; }

/* x^(-1/3) for normal x > 0: bit-trick seed + 5 Newton steps y <- y*(4 - x*y^3)/3 */
static double det_rcbrt(double x)
{
    int64_t i = (int64_t)d2b(x);
    i = INT64_C(0x553ef0ff289dd796) - i / 3;
    double y = b2d((uint64_t)i);
    const double third = 1.0 / 3.0;
    for (int it = 0; it < 5; ++it) {
        double y3 = y * y * y;
        double r = 1.0 - x * y3;
        y = y + y * (r * third);
    }
    return y;
}

/* cube root for normal x > 0: c = x*y*y with y = x^(-1/3), then one correction step */
ORC_API double orc_det_cbrt(double x)
{
    double y = det_rcbrt(x);
    double yy = y * y;
    double c = x * yy;
    double e = c * c * c - x;
    c = c - e * (yy * (1.0 / 3.0));
    return c;
}

/* t^(2.4) for normal t > 0: with y = t^(-1/5), (t*y)^3 = t^(12/5) */
ORC_API double orc_det_pow24(double t)
{
    int64_t i = (int64_t)d2b(t);
    i = INT64_C(0x4cb8a8c154c985f0) - i / 5;
    double y = b2d((uint64_t)i);
    for (int it = 0; it < 5; ++it) {
        double y2 = y * y;
        double y5 = y2 * y2 * y;
        double r = 1.0 - t * y5;
        y = y + y * (r * 0.2);
    }
    double p = t * y;
    return p * p * p;
}

/* ------------------------------------------------------------------------------------------ */
/* SLIC stage (skimage 0.18 semantics)                                                         */
/* ------------------------------------------------------------------------------------------ */

/* skimage.color.rgb2lab on one pixel already in [0,1] float64 (colorconv.py rgb2xyz + xyz2lab,
 * D65 / 2deg white point).  The 3x3 product is evaluated left to right, unfused. */
ORC_API void orc_rgb2lab_px(const double rgb[3], double lab[3])
{
    double lin[3];
    for (int c = 0; c < 3; ++c) {
        double v = rgb[c];
        if (v > 0.04045)
            lin[c] = orc_det_pow24((v + 0.055) / 1.055);
        else
            lin[c] = v / 12.92;
    }
    double X = lin[0] * 0.412453 + lin[1] * 0.357580 + lin[2] * 0.180423;
    double Y = lin[0] * 0.212671 + lin[1] * 0.715160 + lin[2] * 0.072169;
    double Z = lin[0] * 0.019334 + lin[1] * 0.119193 + lin[2] * 0.950227;
    double xyz[3] = { X / 0.95047, Y / 1.0, Z / 1.08883 };
    double f[3];
    for (int c = 0; c < 3; ++c) {
        double t = xyz[c];
        if (t > 0.008856)
            f[c] = orc_det_cbrt(t);
        else
            f[c] = 7.787 * t + 16.0 / 116.0;
    }
    lab[0] = (116.0 * f[1]) - 16.0;
    lab[1] = 500.0 * (f[0] - f[1]);
    lab[2] = 200.0 * (f[1] - f[2]);
}

/* scipy.ndimage 'reflect' boundary (half-sample symmetric: d c b a | a b c d | d c b a) */
static inline int reflect_idx(int i, int n)
{
    if (n == 1) return 0;
    int p = 2 * n;
    i %= p;
    if (i < 0) i += p;
    if (i >= n) i = p - 1 - i;
    return i;
}

/* one scipy correlate1d pass with a symmetric odd kernel (ni_filters.c NI_Correlate1D,
 * symmetric branch), w[0] = centre tap: tmp = x[c]*w[0]; for j = r..1: tmp += (x[c-j] + x[c+j]) * w[j] */
static void blur_axis(const double *src, double *dst, int D, int H, int W, int axis,
                      const double *w, int r)
{
    int dims[3] = { D, H, W };
    long strides[3] = { (long)H * W, W, 1 };
    int n = dims[axis];
    long st = strides[axis];
    for (int z = 0; z < D; ++z)
        for (int y = 0; y < H; ++y)
            for (int x = 0; x < W; ++x) {
                int pos[3] = { z, y, x };
                int c = pos[axis];
                long base = (long)z * strides[0] + (long)y * strides[1] + x - (long)c * st;
                double tmp = src[base + (long)c * st] * w[0];
                for (int j = r; j >= 1; --j) {
                    double a = src[base + (long)reflect_idx(c - j, n) * st];
                    double b = src[base + (long)reflect_idx(c + j, n) * st];
                    tmp += (a + b) * w[j];
                }
                dst[(long)z * strides[0] + (long)y * strides[1] + x] = tmp;
            }
}

/* Gaussian blur of one channel volume along z, y, x in that order (scipy gaussian_filter axis
 * order); an axis with wr < 0 is skipped (sigma == 0).  In place through a scratch buffer. */
ORC_API void orc_gauss_blur3(double *vol, int D, int H, int W,
                             const double *wz, int rz, const double *wy, int ry,
                             const double *wx, int rx)
{
    size_t n = (size_t)D * H * W;
    double *tmp = (double *)malloc(n * sizeof(double));
    if (rz >= 0) { blur_axis(vol, tmp, D, H, W, 0, wz, rz); memcpy(vol, tmp, n * sizeof(double)); }
    if (ry >= 0) { blur_axis(vol, tmp, D, H, W, 1, wy, ry); memcpy(vol, tmp, n * sizeof(double)); }
    if (rx >= 0) { blur_axis(vol, tmp, D, H, W, 2, wx, rx); memcpy(vol, tmp, n * sizeof(double)); }
    free(tmp);
}

/* Pre-processing of superpixels.py:50-54 + slic_superpixels.py (0.18): min-max normalise,
 * rgb2lab, Gaussian blur (sigma/spacing per axis, z then y then x), times 1/compactness.
 *   img   : H*W*3 interleaved, dtype 0 = uint8, 1 = float64
 *   norm  : 1 -> (v - vmin) / (vmax - vmin)   (uint8: integer subtraction first, as numpy does)
 *   out   : planar [3][D=1][H][W] float64 (L plane, a plane, b plane) */
ORC_API void orc_slic_preprocess_color2d(const void *img, int dtype, int H, int W, int norm,
                                         double vmin, double vmax,
                                         const double *wz, int rz, const double *wy, int ry,
                                         const double *wx, int rx, double ratio, double *out)
{
    size_t n = (size_t)H * W;
    const uint8_t *u8 = (const uint8_t *)img;
    const double *f64 = (const double *)img;
    double range = vmax - vmin;
    for (size_t p = 0; p < n; ++p) {
        double rgb[3], lab[3];
        for (int c = 0; c < 3; ++c) {
            double v;
            if (dtype == 0) {
                if (norm) v = (double)(uint8_t)(u8[3 * p + c] - (uint8_t)vmin) / range;
                else v = (double)u8[3 * p + c] * (1.0 / 255);   /* img_as_float(uint8) */
            } else {
                v = f64[3 * p + c];
                if (norm) v = (v - vmin) / range;
            }
            rgb[c] = v;
        }
        orc_rgb2lab_px(rgb, lab);
        for (int c = 0; c < 3; ++c) out[c * n + p] = lab[c];
    }
    for (int c = 0; c < 3; ++c)
        orc_gauss_blur3(out + c * n, 1, H, W, wz, rz, wy, ry, wx, rx);
    for (size_t p = 0; p < 3 * n; ++p) out[p] = out[p] * ratio;
}

/* gray volume pre-processing (superpixels.py:104-106, multichannel=False): no Lab, blur, scale */
ORC_API void orc_slic_preprocess_gray3d(const double *img, int D, int H, int W,
                                        const double *wz, int rz, const double *wy, int ry,
                                        const double *wx, int rx, double ratio, double *out)
{
    size_t n = (size_t)D * H * W;
    memcpy(out, img, n * sizeof(double));
    orc_gauss_blur3(out, D, H, W, wz, rz, wy, ry, wx, rx);
    for (size_t p = 0; p < n; ++p) out[p] = out[p] * ratio;
}

static inline double i64_to_double(int64_t v)
{
    /* (double)(int32 high) * 2^32 + (double)(uint32 low): one rounding, round-to-nearest-even */
    int32_t h = (int32_t)(v >> 32);
    uint32_t l = (uint32_t)(v & 0xffffffff);
    return (double)h * 4294967296.0 + (double)l;
}

/* _slic_cython (skimage/segmentation/_slic.pyx, 0.18) for C channels planar, D x H x W.
 *   centroids : K x (3 + C) row-major [z, y, x, c0..]; colour part must be zero on entry
 *               (slic_superpixels.py concatenates zeros); updated in place
 *   nearest   : D*H*W int32 out (start_label NOT added)
 *   dead centroids (no pixel) keep a NaN position in skimage and never match again; here they
 *   are flagged by count == 0 and skipped (same observable behaviour on x86). */
/* Fixed-point format of the centroid colour sums: value * 2^f truncated toward zero, with
 * f = 46 - e where 2^e > max |pre-processed value| (frexp exponent; f as for 1.0 if the image is all zero).
 * The sum of those integers is exact and order-independent (128-bit accumulator here, int64 partial sums
 * + two int64 limbs on the GPU); it differs from the true sum by less than n * 2^-f, i.e. by less than
 * 2^-44 relative to the value range per pixel -- the size of the rounding noise of the reference's own
 * sequential fp64 accumulation (_slic.pyx adds pixel after pixel into a double). */
ORC_API int orc_fix_bits(const double *img, size_t count)
{
    double m = 0.0;
    for (size_t i = 0; i < count; ++i) {
        double a = fabs(img[i]);
        if (a > m) m = a;
    }
    int e = 1;
    if (m > 0.0) (void)frexp(m, &e);            /* m = frac * 2^e, 0.5 <= frac < 1  ->  m < 2^e */
    return 46 - e;
}

ORC_API void orc_slic_iterate(const double *img, int C, int D, int H, int W, int K,
                              double *centroids, int step_z, int step_y, int step_x, double step,
                              const double spacing[3], int max_iter, int slic_zero, int32_t *nearest)
{
    size_t n = (size_t)D * H * W;
    int F = 3 + C;
    /* SLICO (slic_zero): colour distances are divided by the largest colour distance seen so far inside the
     * segment; "the colors are scaled before ... so max_color_sq can be initialised as all ones" (_slic.pyx) */
    double *max_dist_color = (double *)malloc((size_t)K * sizeof(double));
    for (int k = 0; k < K; ++k) max_dist_color[k] = 1.0;
    double *distance = (double *)malloc(n * sizeof(double));
    int64_t *cnt = (int64_t *)calloc(K, sizeof(int64_t));
    int64_t *csum = (int64_t *)calloc((size_t)K * 3, sizeof(int64_t));
    __int128 *fsum = (__int128 *)calloc((size_t)K * C, sizeof(__int128));
    uint8_t *dead = (uint8_t *)calloc(K, 1);
    double sz = spacing[0], sy = spacing[1], sx = spacing[2];
    double spatial_weight = 1.0 / (step * step);
    const int fbits = orc_fix_bits(img, n * (size_t)C);
    const double fscale = ldexp(1.0, fbits), finv = ldexp(1.0, -fbits);
    for (size_t p = 0; p < n; ++p) nearest[p] = -1;

    for (int it = 0; it < max_iter; ++it) {
        int change = 0;
        for (size_t p = 0; p < n; ++p) distance[p] = DBL_MAX;
        for (int k = 0; k < K; ++k) {
            if (dead[k]) continue;
            const double *seg = centroids + (size_t)k * F;
            double cz = seg[0], cy = seg[1], cx = seg[2];
            double a;
            a = cz - 2 * step_z; long z_min = (long)(a > 0 ? a : 0);
            a = cz + 2 * step_z + 1; long z_max = (long)(a < D ? a : D);
            a = cy - 2 * step_y; long y_min = (long)(a > 0 ? a : 0);
            a = cy + 2 * step_y + 1; long y_max = (long)(a < H ? a : H);
            a = cx - 2 * step_x; long x_min = (long)(a > 0 ? a : 0);
            a = cx + 2 * step_x + 1; long x_max = (long)(a < W ? a : W);
            for (long z = z_min; z < z_max; ++z) {
                double tz = sz * (cz - (double)z);
                double dz = tz * tz;
                for (long y = y_min; y < y_max; ++y) {
                    double ty = sy * (cy - (double)y);
                    double dy = ty * ty;
                    for (long x = x_min; x < x_max; ++x) {
                        size_t p = ((size_t)z * H + y) * W + x;
                        double tx = sx * (cx - (double)x);
                        double dist_center = (dz + dy + tx * tx) * spatial_weight;
                        double dist_color = 0;
                        for (int c = 0; c < C; ++c) {
                            double t = img[(size_t)c * n + p] - seg[3 + c];
                            dist_color += t * t;
                        }
                        if (slic_zero) dist_center += dist_color / max_dist_color[k];
                        else dist_center += dist_color;
                        if (distance[p] > dist_center) {
                            nearest[p] = k;
                            distance[p] = dist_center;
                            change = 1;
                        }
                    }
                }
            }
        }
        if (!change) break;
        /* recompute centres: integer coordinate sums are exact; colour sums are exact fixed point */
        memset(cnt, 0, K * sizeof(int64_t));
        memset(csum, 0, (size_t)K * 3 * sizeof(int64_t));
        memset(fsum, 0, (size_t)K * C * sizeof(__int128));
        for (int z = 0; z < D; ++z)
            for (int y = 0; y < H; ++y)
                for (int x = 0; x < W; ++x) {
                    size_t p = ((size_t)z * H + y) * W + x;
                    int k = nearest[p];
                    if (k < 0) continue;
                    cnt[k] += 1;
                    csum[3 * k + 0] += z; csum[3 * k + 1] += y; csum[3 * k + 2] += x;
                    for (int c = 0; c < C; ++c)
                        fsum[(size_t)k * C + c] += (__int128)(int64_t)trunc(img[(size_t)c * n + p] * fscale);
                }
        for (int k = 0; k < K; ++k) {
            double *seg = centroids + (size_t)k * F;
            if (cnt[k] == 0) {
                dead[k] = 1;
                for (int c = 0; c < F; ++c) seg[c] = NAN;
                continue;
            }
            double nn = (double)cnt[k];
            for (int c = 0; c < 3; ++c) seg[c] = (double)csum[3 * k + c] / nn;
            for (int c = 0; c < C; ++c) {
                /* canonical split  sum = hi * 2^24 + lo,  0 <= lo < 2^24;  both parts are exact doubles */
                __int128 t = fsum[(size_t)k * C + c];
                int64_t hi = (int64_t)(t >> 24), lo = (int64_t)(t & 0xffffff);
                seg[3 + c] = ((i64_to_double(hi) * 16777216.0 + (double)lo) * finv) / nn;
            }
        }
        /* SLICO: update the colour-distance maxima with the distances to the NEW centres; only increases
         * ("The reference implementation seems to only change the color if it increases", _slic.pyx) */
        if (slic_zero)
            for (size_t p = 0; p < n; ++p) {
                int k = nearest[p];
                if (k < 0) continue;
                const double *seg = centroids + (size_t)k * F;
                double dist_color = 0;
                for (int c = 0; c < C; ++c) {
                    double t = img[(size_t)c * n + p] - seg[3 + c];
                    dist_color += t * t;
                }
                if (max_dist_color[k] < dist_color) max_dist_color[k] = dist_color;
            }
    }
    free(max_dist_color);
    free(distance); free(cnt); free(csum); free(fsum); free(dead);
}

/* _enforce_label_connectivity_cython (skimage/segmentation/_slic.pyx, 0.18), literal restatement.
 * segments: labels INCLUDING start_label offset; mask_label = start_label - 1. */
ORC_API void orc_enforce_connectivity(const int32_t *segments, int D, int H, int W,
                                      long min_size, long max_size, int start_label,
                                      int32_t *connected)
{
    static const int ddx[6] = { 1, -1, 0, 0, 0, 0 };
    static const int ddy[6] = { 0, 0, 1, -1, 0, 0 };
    static const int ddz[6] = { 0, 0, 0, 0, 1, -1 };
    size_t n = (size_t)D * H * W;
    int32_t mask_label = start_label - 1;
    for (size_t p = 0; p < n; ++p) connected[p] = mask_label;
    int32_t current_new_label = start_label;
    long cap = max_size > 0 ? max_size : 1;
    int32_t *coord = (int32_t *)malloc((size_t)cap * 3 * sizeof(int32_t));
    for (int z = 0; z < D; ++z)
        for (int y = 0; y < H; ++y)
            for (int x = 0; x < W; ++x) {
                size_t p = ((size_t)z * H + y) * W + x;
                if (segments[p] == mask_label) continue;
                if (connected[p] > mask_label) continue;
                int32_t adjacent = 0;
                int32_t label = segments[p];
                connected[p] = current_new_label;
                long size = 1, visited = 0;
                coord[0] = z; coord[1] = y; coord[2] = x;
                while (visited < size && size < max_size) {
                    for (int i = 0; i < 6; ++i) {
                        int zz = coord[3 * visited + 0] + ddz[i];
                        int yy = coord[3 * visited + 1] + ddy[i];
                        int xx = coord[3 * visited + 2] + ddx[i];
                        if (xx >= 0 && xx < W && yy >= 0 && yy < H && zz >= 0 && zz < D) {
                            size_t q = ((size_t)zz * H + yy) * W + xx;
                            if (segments[q] == label && connected[q] == mask_label) {
                                connected[q] = current_new_label;
                                coord[3 * size + 0] = zz;
                                coord[3 * size + 1] = yy;
                                coord[3 * size + 2] = xx;
                                size += 1;
                                if (size >= max_size) break;
                            } else if (connected[q] > mask_label && connected[q] != current_new_label) {
                                adjacent = connected[q];
                            }
                        }
                    }
                    visited += 1;
                }
                if (size < min_size) {
                    for (long i = 0; i < size; ++i)
                        connected[((size_t)coord[3 * i] * H + coord[3 * i + 1]) * W + coord[3 * i + 2]] = adjacent;
                } else {
                    current_new_label += 1;
                }
            }
    free(coord);
}

/* ------------------------------------------------------------------------------------------ */
/* descriptor stage: features_cython.pyx restated                                              */
/* ------------------------------------------------------------------------------------------ */

/* normColorFeatures, features_cython.pyx:59-78 */
static void norm_color_features(const int32_t *seg, int H, int W, int nb, double *features)
{
    int32_t *count = (int32_t *)calloc(nb, sizeof(int32_t));
    for (int x = 0; x < H; ++x)
        for (int y = 0; y < W; ++y)
            count[seg[(size_t)x * W + y]] += 1;
    for (int z = 0; z < 3; ++z)
        for (int i = 0; i < nb; ++i)
            if (count[i] > 0) features[3 * i + z] = features[3 * i + z] / count[i];
    free(count);
}

static int max_label(const int32_t *seg, size_t n)
{
    int m = seg[0];
    for (size_t i = 1; i < n; ++i) if (seg[i] > m) m = seg[i];
    return m;
}

/* computeColorImage2dMean, features_cython.pyx:81-98; img float32 HxWx3, out (max+1)x3 double */
ORC_API int orc_color2d_mean(const float *img, const int32_t *seg, int H, int W, double *features)
{
    int nb = max_label(seg, (size_t)H * W) + 1;
    memset(features, 0, (size_t)nb * 3 * sizeof(double));
    for (int z = 0; z < 3; ++z)
        for (int x = 0; x < H; ++x)
            for (int y = 0; y < W; ++y)
                features[3 * seg[(size_t)x * W + y] + z] += img[((size_t)x * W + y) * 3 + z];
    norm_color_features(seg, H, W, nb, features);
    return nb;
}

/* computeColorImage2dEnergy, features_cython.pyx:101-119 (val*val evaluated in float32) */
ORC_API int orc_color2d_energy(const float *img, const int32_t *seg, int H, int W, double *features)
{
    int nb = max_label(seg, (size_t)H * W) + 1;
    memset(features, 0, (size_t)nb * 3 * sizeof(double));
    for (int z = 0; z < 3; ++z)
        for (int x = 0; x < H; ++x)
            for (int y = 0; y < W; ++y) {
                volatile float val = img[((size_t)x * W + y) * 3 + z];
                volatile float sq = val * val;
                features[3 * seg[(size_t)x * W + y] + z] += sq;
            }
    norm_color_features(seg, H, W, nb, features);
    return nb;
}

/* computeColorImage2dVariance, features_cython.pyx:122-141 (difference and square in float32) */
ORC_API int orc_color2d_variance(const float *img, const int32_t *seg, int H, int W,
                                 const float *mean, double *features)
{
    int nb = max_label(seg, (size_t)H * W) + 1;
    memset(features, 0, (size_t)nb * 3 * sizeof(double));
    for (int z = 0; z < 3; ++z)
        for (int x = 0; x < H; ++x)
            for (int y = 0; y < W; ++y) {
                int s = seg[(size_t)x * W + y];
                volatile float v = img[((size_t)x * W + y) * 3 + z] - mean[3 * s + z];
                volatile float sq = v * v;
                features[3 * s + z] += sq;
            }
    norm_color_features(seg, H, W, nb, features);
    return nb;
}

/* computeGrayImage3dMean / Energy / Variance, features_cython.pyx:144-219
 * which: 0 mean, 1 energy, 2 variance (mean32 required) */
ORC_API int orc_gray3d_stat(const float *img, const int32_t *seg, int D, int H, int W, int which,
                            const float *mean32, double *features)
{
    size_t n = (size_t)D * H * W;
    int nb = max_label(seg, n) + 1;
    int32_t *count = (int32_t *)calloc(nb, sizeof(int32_t));
    memset(features, 0, (size_t)nb * sizeof(double));
    for (size_t p = 0; p < n; ++p) {
        int idx = seg[p];
        count[idx] += 1;
        if (which == 0) {
            features[idx] += img[p];
        } else if (which == 1) {
            volatile float sq = img[p] * img[p];
            features[idx] += sq;
        } else {
            volatile float v = img[p] - mean32[idx];
            volatile float sq = v * v;
            features[idx] += sq;
        }
    }
    for (int i = 0; i < nb; ++i)
        if (count[i] > 0) features[i] = features[i] / count[i];
    free(count);
    return nb;
}

/* ------------------------------------------------------------------------------------------ */
/* graph stage: superpixels.py:115-242 restated                                                */
/* ------------------------------------------------------------------------------------------ */

static int cmp_i64(const void *a, const void *b)
{
    int64_t x = *(const int64_t *)a, y = *(const int64_t *)b;
    return (x > y) - (x < y);
}

/* make_graph_segm_connect_grid2d_conn4 / 3d_conn6 (superpixels.py:157-202) on a D x H x W
 * label grid (D = 1 for 2D).  vertices_out: sorted unique labels (capacity max+1);
 * edges_out: pairs [a, b] (original ids, a < b) ordered by (b, a) as the hash
 * a' + nb_vertices * b' of superpixels.py:126-130 yields; capacity given by the caller.
 * Returns the number of edges, or -1 if capacity is too small; *nv_out = #vertices. */
ORC_API long orc_adjacency(const int32_t *grid, int D, int H, int W, int32_t *vertices_out,
                           int *nv_out, int32_t *edges_out, long edge_capacity)
{
    size_t n = (size_t)D * H * W;
    int mx = max_label(grid, n);
    int32_t *dense = (int32_t *)malloc(((size_t)mx + 1) * sizeof(int32_t));
    for (int i = 0; i <= mx; ++i) dense[i] = -1;
    for (size_t p = 0; p < n; ++p) dense[grid[p]] = 0;
    int nv = 0;
    for (int i = 0; i <= mx; ++i)
        if (dense[i] == 0) { vertices_out[nv] = i; dense[i] = nv; nv++; }
    *nv_out = nv;
    size_t cap = 1024, m = 0;
    int64_t *hash = (int64_t *)malloc(cap * sizeof(int64_t));
    for (int z = 0; z < D; ++z)
        for (int y = 0; y < H; ++y)
            for (int x = 0; x < W; ++x) {
                size_t p = ((size_t)z * H + y) * W + x;
                int a = dense[grid[p]];
                size_t q[3]; int nq = 0;
                if (x + 1 < W) q[nq++] = p + 1;
                if (y + 1 < H) q[nq++] = p + W;
                if (z + 1 < D) q[nq++] = p + (size_t)H * W;
                for (int j = 0; j < nq; ++j) {
                    int b = dense[grid[q[j]]];
                    if (a == b) continue;
                    int lo = a < b ? a : b, hi = a < b ? b : a;
                    if (m == cap) { cap *= 2; hash = (int64_t *)realloc(hash, cap * sizeof(int64_t)); }
                    hash[m++] = (int64_t)lo + (int64_t)nv * hi;
                }
            }
    qsort(hash, m, sizeof(int64_t), cmp_i64);
    long ne = 0;
    for (size_t i = 0; i < m; ++i) {
        if (i > 0 && hash[i] == hash[i - 1]) continue;
        if (ne >= edge_capacity) { ne = -1; break; }
        edges_out[2 * ne + 0] = vertices_out[hash[i] % nv];
        edges_out[2 * ne + 1] = vertices_out[hash[i] / nv];
        ne++;
    }
    free(hash); free(dense);
    return ne;
}

/* superpixel_centers (superpixels.py:205-242): mean coordinate per label; labels without pixels
 * get -1 in every coordinate.  out: (max+1) x ndim doubles (ndim = 2 if D == 1 and as2d). */
ORC_API int orc_centers(const int32_t *grid, int D, int H, int W, int as2d, double *out)
{
    size_t n = (size_t)D * H * W;
    int nb = max_label(grid, n) + 1;
    int nd = as2d ? 2 : 3;
    int64_t *sum = (int64_t *)calloc((size_t)nb * 3, sizeof(int64_t));
    int64_t *cnt = (int64_t *)calloc(nb, sizeof(int64_t));
    for (int z = 0; z < D; ++z)
        for (int y = 0; y < H; ++y)
            for (int x = 0; x < W; ++x) {
                int k = grid[((size_t)z * H + y) * W + x];
                cnt[k]++; sum[3 * k] += z; sum[3 * k + 1] += y; sum[3 * k + 2] += x;
            }
    for (int k = 0; k < nb; ++k)
        for (int c = 0; c < nd; ++c) {
            int src = as2d ? c + 1 : c;
            out[(size_t)k * nd + c] = cnt[k] ? (double)sum[3 * k + src] / (double)cnt[k] : -1.0;
        }
    free(sum); free(cnt);
    return nb;
}

/* ------------------------------------------------------------------------------------------ */
/* GraphCut stage: gco-wrapper `cut_general_graph(..., algorithm='expansion')` restated        */
/* ------------------------------------------------------------------------------------------ */

/* Dinic max-flow on a small graph with int64 capacities; node 0..n-1, source n, sink n+1. */
typedef struct { int to; int64_t cap; } Arc;
typedef struct {
    int n; int m, mcap; Arc *arcs; int *head, *next; int *level, *iter;
} Flow;

static void flow_init(Flow *g, int n, int mcap)
{
    g->n = n; g->m = 0; g->mcap = mcap;
    g->arcs = (Arc *)malloc((size_t)mcap * sizeof(Arc));
    g->next = (int *)malloc((size_t)mcap * sizeof(int));
    g->head = (int *)malloc((size_t)n * sizeof(int));
    g->level = (int *)malloc((size_t)n * sizeof(int));
    g->iter = (int *)malloc((size_t)n * sizeof(int));
    for (int i = 0; i < n; ++i) g->head[i] = -1;
}
static void flow_free(Flow *g) { free(g->arcs); free(g->next); free(g->head); free(g->level); free(g->iter); }
static void flow_add(Flow *g, int u, int v, int64_t cuv, int64_t cvu)
{
    g->arcs[g->m].to = v; g->arcs[g->m].cap = cuv; g->next[g->m] = g->head[u]; g->head[u] = g->m++;
    g->arcs[g->m].to = u; g->arcs[g->m].cap = cvu; g->next[g->m] = g->head[v]; g->head[v] = g->m++;
}
static int flow_bfs(Flow *g, int s, int t)
{
    int *queue = (int *)malloc((size_t)g->n * sizeof(int));
    for (int i = 0; i < g->n; ++i) g->level[i] = -1;
    int qh = 0, qt = 0; queue[qt++] = s; g->level[s] = 0;
    while (qh < qt) {
        int u = queue[qh++];
        for (int e = g->head[u]; e >= 0; e = g->next[e])
            if (g->arcs[e].cap > 0 && g->level[g->arcs[e].to] < 0) {
                g->level[g->arcs[e].to] = g->level[u] + 1; queue[qt++] = g->arcs[e].to;
            }
    }
    free(queue);
    return g->level[t] >= 0;
}
static int64_t flow_dfs(Flow *g, int u, int t, int64_t f)
{
    if (u == t) return f;
    for (int *e = &g->iter[u]; *e >= 0; *e = g->next[*e]) {
        Arc *a = &g->arcs[*e];
        if (a->cap > 0 && g->level[a->to] == g->level[u] + 1) {
            int64_t d = flow_dfs(g, a->to, t, f < a->cap ? f : a->cap);
            if (d > 0) { a->cap -= d; g->arcs[*e ^ 1].cap += d; return d; }
        }
    }
    return 0;
}
static int64_t flow_max(Flow *g, int s, int t)
{
    int64_t total = 0;
    while (flow_bfs(g, s, t)) {
        for (int i = 0; i < g->n; ++i) g->iter[i] = g->head[i];
        int64_t f;
        while ((f = flow_dfs(g, s, t, INT64_MAX)) > 0) total += f;
    }
    return total;
}

typedef struct {
    int K, C, E;
    const int32_t *edges; const int32_t *w; const int32_t *unary; const int32_t *smooth;
    int32_t *labeling;
} GCO;

static int64_t gco_energy(const GCO *g)
{
    int64_t e = 0;
    for (int i = 0; i < g->K; ++i) e += g->unary[(size_t)i * g->C + g->labeling[i]];
    for (int j = 0; j < g->E; ++j) {
        int a = g->edges[2 * j], b = g->edges[2 * j + 1];
        e += (int64_t)g->w[j] * g->smooth[g->labeling[a] * g->C + g->labeling[b]];
    }
    return e;
}

/* One expansion move (GCoptimization::alpha_expansion of GCO-v3 + Kolmogorov energy.h terms).
 * Binary variable per active site: 0 = take alpha (SOURCE side), 1 = keep (SINK side).
 * Cut convention of maxflow-v3 `what_segment(i, default=SOURCE)`: a site keeps its label only if
 * it belongs to the sink tree at termination == it can still reach the sink in the residual
 * graph; everything else (source tree and free nodes) takes alpha.
 * Returns 1 if the energy strictly decreased (move accepted). */
static int gco_alpha_expansion(GCO *g, int alpha, int64_t *energy)
{
    int K = g->K, C = g->C;
    int *var = (int *)malloc((size_t)K * sizeof(int));
    int size = 0;
    for (int i = 0; i < K; ++i) var[i] = (g->labeling[i] != alpha) ? size++ : -1;
    if (size == 0) { free(var); return 0; }
    int64_t *tsrc = (int64_t *)calloc(size, sizeof(int64_t));   /* cost paid if var = 1 ... source cap */
    int64_t *tsnk = (int64_t *)calloc(size, sizeof(int64_t));
    Flow f; flow_init(&f, size + 2, 2 * (g->E + size * 2) + 16);
    int S = size, T = size + 1;
    /* add_term1(x, E0, E1) == add_tweights(x, E1, E0): source cap E1 (cut when x on SINK side ->
     * hmm: in energy.h a node on the SOURCE side has value 0 and pays the sink-link capacity). */
    for (int i = 0; i < K; ++i) {
        if (var[i] < 0) continue;
        /* E0 = D(alpha) [x = 0], E1 = D(current) [x = 1] */
        tsrc[var[i]] += g->unary[(size_t)i * C + g->labeling[i]];   /* cap_source = E1 */
        tsnk[var[i]] += g->unary[(size_t)i * C + alpha];            /* cap_sink   = E0 */
    }
    for (int j = 0; j < g->E; ++j) {
        int p = g->edges[2 * j], q = g->edges[2 * j + 1];
        int64_t w = g->w[j];
        int lp = g->labeling[p], lq = g->labeling[q];
        if (var[p] < 0 && var[q] < 0) continue;
        if (var[p] >= 0 && var[q] < 0) {
            /* add_term1(p, V(alpha, lq = alpha), V(lp, lq)) */
            tsnk[var[p]] += w * g->smooth[alpha * C + lq];
            tsrc[var[p]] += w * g->smooth[lp * C + lq];
        } else if (var[p] < 0 && var[q] >= 0) {
            tsnk[var[q]] += w * g->smooth[lp * C + alpha];
            tsrc[var[q]] += w * g->smooth[lp * C + lq];
        } else {
            /* add_term2(x, y, A = E00, B = E01, C = E10, D = E11) with x = var[p], y = var[q] */
            int x = var[p], y = var[q];
            int64_t A = w * g->smooth[alpha * C + alpha];
            int64_t B = w * g->smooth[alpha * C + lq];
            int64_t Cc = w * g->smooth[lp * C + alpha];
            int64_t Dd = w * g->smooth[lp * C + lq];
            /* add_tweights(x, D, A) */
            tsrc[x] += Dd; tsnk[x] += A;
            B -= A; Cc -= Dd;
            if (B < 0) {
                tsnk[x] += B;  /* add_tweights(x, 0, B) */
                tsnk[y] += -B; /* add_tweights(y, 0, -B) */
                flow_add(&f, x, y, 0, B + Cc);
            } else if (Cc < 0) {
                tsnk[x] += -Cc; /* add_tweights(x, 0, -C) */
                tsnk[y] += Cc;  /* add_tweights(y, 0, C) */
                flow_add(&f, x, y, B + Cc, 0);
            } else {
                flow_add(&f, x, y, B, Cc);
            }
        }
    }
    /* maxflow-v3 add_tweights keeps only the difference source-sink per node; the common part is
     * constant flow.  Equivalent: add both t-links (negative values shifted by the node minimum). */
    for (int i = 0; i < size; ++i) {
        int64_t mn = tsrc[i] < tsnk[i] ? tsrc[i] : tsnk[i];
        flow_add(&f, S, i, tsrc[i] - mn, 0);
        flow_add(&f, i, T, tsnk[i] - mn, 0);
    }
    flow_max(&f, S, T);
    /* sink side = nodes that can reach T in the residual graph (reverse BFS from T) */
    uint8_t *sink = (uint8_t *)calloc(size + 2, 1);
    int *queue = (int *)malloc((size_t)(size + 2) * sizeof(int));
    int qh = 0, qt = 0; queue[qt++] = T; sink[T] = 1;
    while (qh < qt) {
        int u = queue[qh++];
        for (int e = f.head[u]; e >= 0; e = f.next[e]) {
            int v = f.arcs[e].to;
            /* arc v->u is the twin e^1; residual capacity of v->u */
            if (!sink[v] && f.arcs[e ^ 1].cap > 0) { sink[v] = 1; queue[qt++] = v; }
        }
    }
    int32_t *backup = (int32_t *)malloc((size_t)K * sizeof(int32_t));
    memcpy(backup, g->labeling, (size_t)K * sizeof(int32_t));
    for (int i = 0; i < K; ++i)
        if (var[i] >= 0 && !sink[var[i]]) g->labeling[i] = alpha;
    int64_t after = gco_energy(g);
    int accepted = after < *energy;
    if (accepted) *energy = after; else memcpy(g->labeling, backup, (size_t)K * sizeof(int32_t));
    free(backup); free(queue); free(sink); flow_free(&f); free(tsrc); free(tsnk); free(var);
    return accepted;
}

/* integer-energy alpha expansion, GCoptimization::expansion(max_num_iterations) of GCO-v3:
 *   n_iter == -1 : adaptive cycles over the label queue (fixed label order 0..C-1)
 *   n_iter  >  0 : at most n_iter full sweeps, stop when a sweep does not lower the energy
 * labels_out starts from all zeros (GCO default labelling). Returns the final energy. */
ORC_API int64_t orc_alpha_expansion_int(const int32_t *edges, int E, const int32_t *w,
                                        const int32_t *unary, int K, int C, const int32_t *smooth,
                                        int n_iter, int32_t *labels_out)
{
    GCO g = { K, C, E, edges, w, unary, smooth, labels_out };
    for (int i = 0; i < K; ++i) labels_out[i] = 0;
    if (E == 0) {
        /* solveSpecialCases: data costs only -> independent argmin (first minimum) */
        for (int i = 0; i < K; ++i) {
            int best = 0;
            for (int l = 1; l < C; ++l)
                if (unary[(size_t)i * C + l] < unary[(size_t)i * C + best]) best = l;
            labels_out[i] = best;
        }
        return gco_energy(&g);
    }
    int64_t energy = gco_energy(&g);
    int *table = (int *)malloc((size_t)C * sizeof(int));
    for (int l = 0; l < C; ++l) table[l] = l;
    if (n_iter == -1) {
        int *queue_sizes = (int *)malloc((size_t)(C + 2) * sizeof(int));
        int nq = 0; queue_sizes[nq++] = C;
        int next = 0;
        do {
            int queue_size = queue_sizes[nq - 1];
            int start = next;
            do {
                if (!gco_alpha_expansion(&g, table[next], &energy)) {
                    --queue_size;
                    int t = table[next]; table[next] = table[queue_size]; table[queue_size] = t;
                } else {
                    ++next;
                }
            } while (next < queue_size);
            if (next == start) {
                next = queue_sizes[nq - 1];
                nq--;
            } else if (queue_size < queue_sizes[nq - 1] / 2) {
                next = 0;
                queue_sizes[nq++] = queue_size;
            } else {
                next = 0;
            }
        } while (nq > 0);
        free(queue_sizes);
    } else {
        for (int cycle = 0; cycle < n_iter; ++cycle) {
            int64_t before = energy;
            for (int l = 0; l < C; ++l) gco_alpha_expansion(&g, table[l], &energy);
            if (!(energy < before)) break;
        }
    }
    free(table);
    return energy;
}

/* pyGCO float front end (gco/pygco.py cut_general_graph + GCO._convert_*): scale to integers
 *   dwf = max(|unary|max, |w|max * pairwise.max()) + 1e-10
 *   unary_i = trunc(unary / dwf * 100000), w_i = trunc(w / dwf * 1000), smooth_i = trunc(pw * 100) */
ORC_API int64_t orc_cut_general_graph(const int32_t *edges, int E, const double *edge_weights,
                                      const double *unary, int K, int C, const double *pairwise,
                                      int n_iter, int32_t *labels_out)
{
    double mu = 0, mw = 0, mp = -DBL_MAX;
    for (size_t i = 0; i < (size_t)K * C; ++i) if (fabs(unary[i]) > mu) mu = fabs(unary[i]);
    for (int i = 0; i < E; ++i) if (fabs(edge_weights[i]) > mw) mw = fabs(edge_weights[i]);
    for (int i = 0; i < C * C; ++i) if (pairwise[i] > mp) mp = pairwise[i];
    double dwf = (E > 0 && mw * mp > mu ? mw * mp : mu) + 1e-10;
    int32_t *ui = (int32_t *)malloc((size_t)K * C * sizeof(int32_t));
    int32_t *wi = (int32_t *)malloc((size_t)(E > 0 ? E : 1) * sizeof(int32_t));
    int32_t *si = (int32_t *)malloc((size_t)C * C * sizeof(int32_t));
    for (size_t i = 0; i < (size_t)K * C; ++i) ui[i] = (int32_t)((unary[i] / dwf) * 100000);
    for (int i = 0; i < E; ++i) wi[i] = (int32_t)((edge_weights[i] / dwf) * 1000);
    for (int i = 0; i < C * C; ++i) si[i] = (int32_t)(pairwise[i] * 100);
    int64_t e = orc_alpha_expansion_int(edges, E, wi, ui, K, C, si, n_iter, labels_out);
    free(ui); free(wi); free(si);
    return e;
}

/* final LUT gathers, pipelines.py:104,109 */
ORC_API void orc_gather_i32(const int32_t *lut, const int32_t *idx, size_t n, int32_t *out)
{
    for (size_t i = 0; i < n; ++i) out[i] = lut[idx[i]];
}
ORC_API void orc_gather_f64(const double *lut, int C, const int32_t *idx, size_t n, double *out)
{
    for (size_t i = 0; i < n; ++i)
        for (int c = 0; c < C; ++c) out[i * C + c] = lut[(size_t)idx[i] * C + c];
}

/* skimage.measure.label(label_image) with its defaults (background = 0, full connectivity), as called
 * at /root/reference/imsegm/superpixels.py:111 (skimage/measure/_ccomp.pyx): components of equal
 * non-zero value under 8- (2-D) / 26- (3-D) connectivity, numbered 1, 2, ... in raster order of
 * their first element; zeros stay 0.  Returns the number of components. */
static int lcc_find(int32_t *parent, int a)
{
    while (parent[a] != a) {
        parent[a] = parent[parent[a]];
        a = parent[a];
    }
    return a;
}
ORC_API int orc_label_cc(const int32_t *in, int D, int H, int W, int32_t *out)
{
    size_t n = (size_t)D * H * W;
    int32_t *parent = (int32_t *)malloc(n * sizeof(int32_t));
    for (size_t p = 0; p < n; ++p) parent[p] = (int32_t)p;
    for (int z = 0; z < D; ++z)
        for (int y = 0; y < H; ++y)
            for (int x = 0; x < W; ++x) {
                size_t p = ((size_t)z * H + y) * W + x;
                if (in[p] == 0) continue;
                for (int dz = -1; dz <= 0; ++dz)
                    for (int dy = -1; dy <= 1; ++dy)
                        for (int dx = -1; dx <= 1; ++dx) {
                            if (dz == 0 && (dy > 0 || (dy == 0 && dx >= 0))) continue;
                            int zz = z + dz, yy = y + dy, xx = x + dx;
                            if (zz < 0 || yy < 0 || yy >= H || xx < 0 || xx >= W) continue;
                            size_t q = ((size_t)zz * H + yy) * W + xx;
                            if (in[q] != in[p]) continue;
                            int a = lcc_find(parent, (int)p), b = lcc_find(parent, (int)q);
                            if (a < b) parent[b] = a; else if (b < a) parent[a] = b;
                        }
            }
    int count = 0;
    for (size_t p = 0; p < n; ++p) {
        if (in[p] == 0) { out[p] = 0; continue; }
        int r = lcc_find(parent, (int)p);
        if ((size_t)r == p) out[p] = ++count;     /* roots are the raster-first elements */
        else out[p] = out[r];
    }
    free(parent);
    return count;
}

/* imsegm/labeling.py:208-247 histogram_regions_labels_counts: matrix_hist[slic[i], segm[i]] += 1 over the flat
 * arrays (:244-245).  hist: [K][nb], zeroed here; pairs outside the matrix are not counted (the Python code
 * cannot meet them: K = max(slic) + 1, nb = max(segm) + 1, negatives rejected at :236). */
ORC_API void orc_label_hist(const int32_t *slic, const int32_t *annot, size_t n, int K, int nb, int64_t *hist)
{
    memset(hist, 0, (size_t)K * nb * sizeof(int64_t));
    for (size_t i = 0; i < n; ++i) {
        int k = slic[i], a = annot[i];
        if (k < 0 || k >= K || a < 0 || a >= nb) continue;
        hist[(size_t)k * nb + a] += 1;
    }
}

/* ---------------------------------------------------------------------------------------------------------
 * float32 volumes.  scikit-image 0.18 keeps a float32 input in float32 from end to end (slic_superpixels.py:
 * `dtype = image.dtype` ... `_slic_cython[float32]`): the Gaussian filter stores float32 after every axis pass
 * (scipy computes each line in double), `image * ratio` is a float32 product, centroids, spacing, distances and
 * the raster-order running sums of the centroid update are float32.  This is the restatement of that variant
 * for the reference's 3-D call (superpixels.py:104-106, multichannel=False); it is checked against the real
 * scikit-image 0.18.3 in tests/test_golden_skimage.py.  The HIP path does not have it yet (it widens float32
 * volumes to float64): the sequential float32 sums need an order-preserving reduction.
 * --------------------------------------------------------------------------------------------------------- */
static void blur_axis_f32(float *vol, int D, int H, int W, int axis, const double *w, int r)
{
    size_t n = (size_t)D * H * W;
    double *src = (double *)calloc(n, sizeof(double)), *dst = (double *)calloc(n, sizeof(double));
    for (size_t p = 0; p < n; ++p) src[p] = (double)vol[p];
    blur_axis(src, dst, D, H, W, axis, w, r);
    for (size_t p = 0; p < n; ++p) vol[p] = (float)dst[p];          /* scipy casts the line to the output dtype */
    free(src); free(dst);
}

ORC_API void orc_slic_gray3d_f32(const float *img, int D, int H, int W,
                                 const double *wz, int rz, const double *wy, int ry, const double *wx, int rx,
                                 double ratio, int K, const double *centroids_zyx, int step_z, int step_y, int step_x,
                                 float step, const double spacing_in[3], int max_iter,
                                 float *pre_out, int32_t *nearest)
{
    size_t n = (size_t)D * H * W;
    float *im = pre_out;
    memcpy(im, img, n * sizeof(float));
    if (rz >= 0) blur_axis_f32(im, D, H, W, 0, wz, rz);
    if (ry >= 0) blur_axis_f32(im, D, H, W, 1, wy, ry);
    if (rx >= 0) blur_axis_f32(im, D, H, W, 2, wx, rx);
    const float fratio = (float)ratio;                               /* numpy: float32 array * Python float */
    for (size_t p = 0; p < n; ++p) im[p] = im[p] * fratio;

    float *seg = (float *)calloc((size_t)K * 4, sizeof(float));
    for (int k = 0; k < K; ++k)
        for (int c = 0; c < 3; ++c) seg[4 * k + c] = (float)centroids_zyx[3 * k + c];
    float *distance = (float *)malloc(n * sizeof(float));
    long *cnt = (long *)calloc(K, sizeof(long));
    uint8_t *dead = (uint8_t *)calloc(K, 1);
    const float sz = (float)spacing_in[0], sy = (float)spacing_in[1], sx = (float)spacing_in[2];
    const float spatial_weight = (float)(1.0 / ((double)step * (double)step));
    for (size_t p = 0; p < n; ++p) nearest[p] = -1;
    for (int it = 0; it < max_iter; ++it) {
        int change = 0;
        for (size_t p = 0; p < n; ++p) distance[p] = INFINITY;        /* DBL_MAX stored as float32 */
        for (int k = 0; k < K; ++k) {
            if (dead[k]) continue;
            const float cz = seg[4 * k], cy = seg[4 * k + 1], cx = seg[4 * k + 2], cv = seg[4 * k + 3];
            float a;
            a = cz - 2 * step_z; long z_min = (long)(a > 0 ? a : 0);
            a = cz + 2 * step_z + 1; long z_max = (long)(a < D ? a : D);
            a = cy - 2 * step_y; long y_min = (long)(a > 0 ? a : 0);
            a = cy + 2 * step_y + 1; long y_max = (long)(a < H ? a : H);
            a = cx - 2 * step_x; long x_min = (long)(a > 0 ? a : 0);
            a = cx + 2 * step_x + 1; long x_max = (long)(a < W ? a : W);
            for (long z = z_min; z < z_max; ++z) {
                const float tz = sz * (cz - (float)z);
                const float dz = tz * tz;
                for (long y = y_min; y < y_max; ++y) {
                    const float ty = sy * (cy - (float)y);
                    const float dy = ty * ty;
                    for (long x = x_min; x < x_max; ++x) {
                        size_t p = ((size_t)z * H + y) * W + x;
                        const float tx = sx * (cx - (float)x);
                        float dist_center = ((dz + dy) + tx * tx) * spatial_weight;
                        const float t = im[p] - cv;
                        float dist_color = 0.f;
                        dist_color += t * t;
                        dist_center += dist_color;
                        if (distance[p] > dist_center) {
                            nearest[p] = k;
                            distance[p] = dist_center;
                            change = 1;
                        }
                    }
                }
            }
        }
        if (!change) break;
        memset(cnt, 0, K * sizeof(long));
        for (int k = 0; k < K; ++k)
            if (!dead[k]) seg[4 * k] = seg[4 * k + 1] = seg[4 * k + 2] = seg[4 * k + 3] = 0.f;
        for (int z = 0; z < D; ++z)
            for (int y = 0; y < H; ++y)
                for (int x = 0; x < W; ++x) {
                    size_t p = ((size_t)z * H + y) * W + x;
                    int k = nearest[p];
                    if (k < 0) continue;
                    cnt[k] += 1;
                    seg[4 * k] += (float)z;                               /* running float32 sums, raster order */
                    seg[4 * k + 1] += (float)y;
                    seg[4 * k + 2] += (float)x;
                    seg[4 * k + 3] += im[p];
                }
        for (int k = 0; k < K; ++k) {
            if (dead[k]) continue;
            if (cnt[k] == 0) {
                dead[k] = 1;
                continue;
            }
            for (int c = 0; c < 4; ++c) seg[4 * k + c] = seg[4 * k + c] / (float)cnt[k];
        }
    }
    free(seg); free(distance); free(cnt); free(dead);
}
