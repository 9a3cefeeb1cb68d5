// volume.hip -- 3-D gray supervoxel path (BASELINE config 5).
//
// Replaces, for D x H x W gray volumes:
//   skimage.segmentation.slic(im, n_segments, compactness, multichannel=False, spacing=space, sigma=1)
//       as called at /root/reference/imsegm/superpixels.py:104-106   (no Lab, anisotropic spacing)
//   skimage.measure.label(slic_segments)                     superpixels.py:111 (full connectivity, 0 = background)
//   make_graph_segm_connect_grid3d_conn6 / superpixel_centers (3-D branch)   superpixels.py:180-242
// Arithmetic contract of the assignment: identical to oracle orc_slic_iterate (fp64, operation order
// of _slic.pyx with spacing):  d = ((sz*(cz-z))^2 + (sy*(cy-y))^2 + (sx*(cx-x))^2) * (1/step^2) + (v - cv)^2.
//
// Candidate search: the centroids enter their index into the lists of the 64 x 16 x 16 bricks their search window meets
// (k_vol_scatter); an assignment workgroup -- one brick cross-section -- reads the list of its brick only.  Sized for the
// 10^9 voxels / 3 * 10^5 supervoxels of config 5; a float32 volume runs in float32 from end to end, as scikit-image 0.18 does.
#include "slic.h"
#include <hip/hip_ext.h>

namespace imsegm {

__device__ __forceinline__ int vreflect(int i, int n)
{
    if (n == 1) return 0;
    int p = 2 * n;
    i %= p;
    if (i < 0) i += p;
    if (i >= n) i = p - 1 - i;
    return i;
}

template <typename T>
__global__ void __launch_bounds__(256)
k_vol_to_f64(const T *__restrict__ src, size_t n, double off, double scale, double *__restrict__ dst)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    // skimage.img_as_float as an affine map: (v + off) * scale   (uint8: off 0, scale 1 / 255)
    if (i < n) dst[i] = ((double)src[i] + off) * scale;
}

// one scipy correlate1d pass (symmetric taps, 'reflect') along z (0), y (1) or x (2); optional final scale
template <int AXIS>
__global__ void __launch_bounds__(256)
k_vol_blur(const double *__restrict__ src, double *__restrict__ dst, int D, int H, int W, Taps t, double ratio, int scale)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t n = (size_t)D * H * W;
    if (i >= n) return;
    int x = (int)(i % W), y = (int)((i / W) % H), z = (int)(i / ((size_t)W * H));
    double v;
    if (t.r < 0) {
        v = src[i];
    } else {
        v = src[i] * t.w[0];
        for (int j = t.r; j >= 1; --j) {
            size_t a, b;
            if (AXIS == 0) {
                a = ((size_t)vreflect(z - j, D) * H + y) * W + x;
                b = ((size_t)vreflect(z + j, D) * H + y) * W + x;
            } else if (AXIS == 1) {
                a = ((size_t)z * H + vreflect(y - j, H)) * W + x;
                b = ((size_t)z * H + vreflect(y + j, H)) * W + x;
            } else {
                a = ((size_t)z * H + y) * W + vreflect(x - j, W);
                b = ((size_t)z * H + y) * W + vreflect(x + j, W);
            }
            v += (src[a] + src[b]) * t.w[j];
        }
    }
    if (scale) v = v * ratio;
    dst[i] = v;
}

int launch_vol_preprocess(const void *src, int dtype, double off, double scale, int D, int H, int W, const Taps &tz, const Taps &ty,
                          const Taps &tx, double ratio, double *bufA, double *bufB, hipStream_t st)
{
    size_t n = (size_t)D * H * W;
    int grid = cdiv((long)n, 256);
    if (dtype == DT_U8) hipLaunchKernelGGL(k_vol_to_f64<uint8_t>, grid, 256, 0, st, (const uint8_t *)src, n, off, scale, bufA);
    else if (dtype == DT_F32) hipLaunchKernelGGL(k_vol_to_f64<float>, grid, 256, 0, st, (const float *)src, n, off, scale, bufA);
    else hipLaunchKernelGGL(k_vol_to_f64<double>, grid, 256, 0, st, (const double *)src, n, off, scale, bufA);
    hipLaunchKernelGGL(k_vol_blur<0>, grid, 256, 0, st, bufA, bufB, D, H, W, tz, ratio, 0);
    hipLaunchKernelGGL(k_vol_blur<1>, grid, 256, 0, st, bufB, bufA, D, H, W, ty, ratio, 0);
    hipLaunchKernelGGL(k_vol_blur<2>, grid, 256, 0, st, bufA, bufB, D, H, W, tx, ratio, 1);
    HIP_TRY(hipGetLastError());
    return 0;   // result in bufB
}

// ---- centroid table ----------------------------------------------------------------------------------
__device__ __forceinline__ void vol_window(const VolState &s, double cz, double cy, double cx, int *w)
{
    double a;
    a = cz - (double)(2 * s.step_z); w[0] = (int)(a > 0 ? a : 0.0);
    a = cz + (double)(2 * s.step_z); a = a + 1.0; w[1] = (int)(a < (double)s.D ? a : (double)s.D);
    a = cy - (double)(2 * s.step_y); w[2] = (int)(a > 0 ? a : 0.0);
    a = cy + (double)(2 * s.step_y); a = a + 1.0; w[3] = (int)(a < (double)s.H ? a : (double)s.H);
    a = cx - (double)(2 * s.step_x); w[4] = (int)(a > 0 ? a : 0.0);
    a = cx + (double)(2 * s.step_x); a = a + 1.0; w[5] = (int)(a < (double)s.W ? a : (double)s.W);
}

__global__ void k_vol_centroid_init(VolState s)
{
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= s.K) return;
    int ix = k % s.grid_n[2], iy = (k / s.grid_n[2]) % s.grid_n[1], iz = k / (s.grid_n[2] * s.grid_n[1]);
    double cz = (double)(s.grid_0[0] + iz * s.grid_d[0]);
    double cy = (double)(s.grid_0[1] + iy * s.grid_d[1]);
    double cx = (double)(s.grid_0[2] + ix * s.grid_d[2]);
    s.cen[(size_t)k * 4 + 0] = cz;
    s.cen[(size_t)k * 4 + 1] = cy;
    s.cen[(size_t)k * 4 + 2] = cx;
    s.cen[(size_t)k * 4 + 3] = 0.0;
    vol_window(s, cz, cy, cx, s.win + (size_t)k * 6);
    for (int j = 0; j < 6; ++j) s.acc[(size_t)k * 6 + j] = 0;
}

__global__ void k_vol_centroid_finalize(VolState s)
{
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= s.K) return;
    long long *a = s.acc + (size_t)k * 6;
    long long n = a[0];
    int *w = s.win + (size_t)k * 6;
    if (n == 0) {
        for (int j = 0; j < 6; ++j) w[j] = 0;
    } else {
        double nn = (double)n;
        double cz = i64_to_double(a[1]) / nn, cy = i64_to_double(a[2]) / nn, cx = i64_to_double(a[3]) / nn;
        s.cen[(size_t)k * 4 + 0] = cz;
        s.cen[(size_t)k * 4 + 1] = cy;
        s.cen[(size_t)k * 4 + 2] = cx;
        s.cen[(size_t)k * 4 + 3] = fix_value(a[4], a[5], ldexp(1.0, -fix_bits_of(*s.premax))) / nn;
        vol_window(s, cz, cy, cx, w);
    }
    for (int j = 0; j < 6; ++j) a[j] = 0;
}

// every centroid appends itself to the list of each brick its search window meets (order irrelevant: the
// assignment breaks ties on the centroid index explicitly)
__global__ void __launch_bounds__(256) k_vol_scatter(VolState s)
{
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= s.K) return;
    const int *w = s.win + (size_t)k * 6;
    if (w[1] <= w[0] || w[3] <= w[2] || w[5] <= w[4]) return;       // dead centroid: empty window
    const int bz0 = w[0] / VOL_BZ, bz1 = (w[1] - 1) / VOL_BZ;
    const int by0 = w[2] / VOL_BY, by1 = (w[3] - 1) / VOL_BY;
    const int bx0 = w[4] / VOL_BX, bx1 = (w[5] - 1) / VOL_BX;
    for (int bz = bz0; bz <= bz1; ++bz)
        for (int by = by0; by <= by1; ++by)
            for (int bx = bx0; bx <= bx1; ++bx) {
                const int b = (bz * s.nby + by) * s.nbx + bx;
                const int pos = atomicAdd(&s.brick_count[b], 1);
                if (pos < s.brick_cap) s.brick_list[(size_t)b * s.brick_cap + pos] = k;
            }
}

// ---- assignment -----------------------------------------------------------------------------------------
// One wave = 64 consecutive x, VROWS rows of one z slice; a workgroup = four such strips = the cross-section of a brick.
// The workgroup scans the candidate list of its brick (or, if that list overflowed, the whole centroid table) once,
// keeps the windows that meet the cross-section in an LDS list (ballot compaction), and every wave evaluates the
// list for its strip; equal distances go to the lower centroid index, which is what the ascending scan with a
// strict '>' of _slic.pyx yields (the order of evaluation is free: the comparison carries the index).
constexpr int VROWS = 4;

// (round 3: like the float32 kernel below -- the four waves of a workgroup scan the brick's list once, together, stage the records of
// the windows that meet the 64 x 16 cross-section in LDS, and each wave walks them in ascending order of their lower bound over
// its strip, stopping at the first bound above the worst best distance of the strip)
struct VolRec64 {
    double cz, cy, cx, cv;
    int wy0, wy1, wx0, wx1;
};
constexpr int VLIST64 = 512;

template <bool ACCUM>
__global__ void __launch_bounds__(256)
k_vol_assign(VolState s, const double *__restrict__ vol, int32_t *__restrict__ labels)
{
    __shared__ int list[VLIST64];
    __shared__ VolRec64 rec[VLIST64];
    __shared__ int wave_base[4];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rows_per_block = 4 * VROWS;
    const int yb = cdiv(s.H, rows_per_block);
    const int z = blockIdx.y / yb;
    const int Y0 = (blockIdx.y % yb) * rows_per_block, Y1 = min(Y0 + rows_per_block, s.H);
    const int y0 = Y0 + wave * VROWS;
    const int x = blockIdx.x * 64 + lane;
    const int x0w = blockIdx.x * 64, x1w = min(x0w + 64, s.W);
    const bool alive = y0 < s.H;                              // (a wave below the volume takes part in the barriers only)
    const int y1w = min(y0 + VROWS, s.H);
    const bool xin = x < s.W;
    double pv[VROWS], best_d[VROWS];
    int best_k[VROWS];
#pragma unroll
    for (int r = 0; r < VROWS; ++r) {
        bool ok = alive && xin && (y0 + r) < s.H;
        pv[r] = vol[ok ? ((size_t)z * s.H + y0 + r) * s.W + x : 0];
        best_d[r] = DBL_MAX;
        best_k[r] = -1;
    }
    const double fz = (double)z, fx = (double)x;
    int count = 0;                                         // (uniform over the workgroup)
    double wave_worst = DBL_MAX;          // >= the current best distance of every voxel of this wave
    static_assert(VOL_BX == 64 && VOL_BY == 4 * VROWS, "a workgroup lies inside one brick");
    const int brick = ((z / VOL_BZ) * s.nby + (Y0 / VOL_BY)) * s.nbx + blockIdx.x;
    const int bcount = s.brick_count[brick];
    const bool whole = bcount > s.brick_cap;                   // list overflow: scan every centroid
    const int *__restrict__ blist = s.brick_list + (size_t)brick * s.brick_cap;
    const int nscan = whole ? s.K : bcount;
    const int nblk = cdiv(nscan, 256);
    constexpr int PER = VLIST64 / 64;
    for (int b = 0; b < nblk; ++b) {
        const int i = b * 256 + tid;
        int k = 0;
        bool hit = false;
        if (i < nscan) {
            k = whole ? i : blist[i];
            const int *w = s.win + (size_t)k * 6;
            hit = z >= w[0] && z < w[1] && w[2] < Y1 && w[3] > Y0 && w[4] < x1w && w[5] > x0w;
        }
        const unsigned long long m = __ballot(hit);
        if (lane == 0) wave_base[wave] = __popcll(m);
        __syncthreads();
        int base = count;
        for (int w2 = 0; w2 < wave; ++w2) base += wave_base[w2];
        const int added = wave_base[0] + wave_base[1] + wave_base[2] + wave_base[3];
        if (hit) list[base + __popcll(m & ((1ULL << lane) - 1ULL))] = k;
        count += added;
        __syncthreads();
        if (count <= VLIST64 - 256 && b + 1 < nblk) continue;
        for (int c = tid; c < count; c += 256) {
            const int ck = list[c];
            const int *w = s.win + (size_t)ck * 6;
            VolRec64 rc;
            rc.cz = s.cen[(size_t)ck * 4]; rc.cy = s.cen[(size_t)ck * 4 + 1]; rc.cx = s.cen[(size_t)ck * 4 + 2];
            rc.cv = s.cen[(size_t)ck * 4 + 3];
            rc.wy0 = w[2]; rc.wy1 = w[3]; rc.wx0 = w[4]; rc.wx1 = w[5];
            rec[c] = rc;
        }
        __syncthreads();
        if (alive) {
            // Exact pruning: the spatial part of the distance to the nearest point of the strip is a lower bound of the
            // distance of every voxel of the strip (every operation is monotone in IEEE arithmetic and the colour term
            // is >= 0); a candidate whose bound exceeds the worst current best of the wave can neither win nor tie.
            double lbl[PER];
#pragma unroll
            for (int j = 0; j < PER; ++j) {
                const int c = lane + 64 * j;
                lbl[j] = DBL_MAX;
                if (c < count) {
                    const VolRec64 rc = rec[c];
                    if (rc.wy0 < y1w && rc.wy1 > y0) {
                        const double tz = s.sz * (rc.cz - fz);
                        const double dz = tz * tz;
                        const double yn = fmin(fmax(rc.cy, (double)y0), (double)(y1w - 1));
                        const double xn = fmin(fmax(rc.cx, (double)x0w), (double)(x1w - 1));
                        const double tyl = s.sy * (rc.cy - yn), txl = s.sx * (rc.cx - xn);
                        lbl[j] = (dz + tyl * tyl + txl * txl) * s.spatial_weight;
                    }
                }
            }
            int since_refresh = 0;
            while (true) {
                double mine = lbl[0];
#pragma unroll
                for (int j = 1; j < PER; ++j) mine = fmin(mine, lbl[j]);
                const double wm = wave_min_f64(mine);
                if (!(wm <= wave_worst) || wm == DBL_MAX) break;   // (DBL_MAX: nothing left in the batch)
                const unsigned long long own = __ballot(mine == wm);
                const int src = __ffsll((long long)own) - 1;
                int jsel = 0;
#pragma unroll
                for (int j = PER - 1; j >= 0; --j)
                    if (lbl[j] == wm) jsel = j;
                jsel = __builtin_amdgcn_readlane(jsel, src);
#pragma unroll
                for (int j = 0; j < PER; ++j)
                    if (lane == src && j == jsel) lbl[j] = DBL_MAX;
                const int c = src + 64 * jsel;
                const int ck = list[c];
                const VolRec64 rc = rec[c];
                const double tz = s.sz * (rc.cz - fz);
                const double dz = tz * tz;
                const bool inx = x >= rc.wx0 && x < rc.wx1;
                const double tx = s.sx * (rc.cx - fx);
                const double dx2 = tx * tx;
#pragma unroll
                for (int r = 0; r < VROWS; ++r) {
                    const int y = y0 + r;
                    if (y < rc.wy0 || y >= rc.wy1) continue;
                    const double ty = s.sy * (rc.cy - (double)y);
                    const double dy = ty * ty;
                    double d = (dz + dy + dx2) * s.spatial_weight;
                    const double t = pv[r] - rc.cv;
                    d = d + t * t;
                    const bool take = inx && (best_d[r] > d || (best_d[r] == d && ck < best_k[r]));
                    best_d[r] = take ? d : best_d[r];               // (selects, no change of the exec mask)
                    best_k[r] = take ? ck : best_k[r];
                }
                if (++since_refresh == 2) {                    // refresh the bound
                    since_refresh = 0;
                    double m2 = 0.0;
#pragma unroll
                    for (int r = 0; r < VROWS; ++r)
                        if (xin && (y0 + r) < s.H) m2 = fmax(m2, best_d[r]);
                    wave_worst = wave_max_f64(m2);
                }
            }
        }
        count = 0;
        __syncthreads();
    }
    if (!alive) return;
    // labels (an uncovered voxel keeps its previous assignment) + accumulation
    unsigned pending = 0;
#pragma unroll
    for (int r = 0; r < VROWS; ++r) {
        if (!(xin && (y0 + r) < s.H)) continue;
        size_t p = ((size_t)z * s.H + y0 + r) * s.W + x;
        if (best_k[r] >= 0) labels[p] = best_k[r];
        else best_k[r] = labels[p];
        if (best_k[r] >= 0) pending |= 1u << r;
    }
    if (!ACCUM) return;
    const double fscale = ldexp(1.0, fix_bits_of(*s.premax));
    while (true) {
        int first = -1;
#pragma unroll
        for (int r = VROWS - 1; r >= 0; --r)
            if (pending & (1u << r)) first = best_k[r];
        unsigned long long vote = __ballot(first >= 0);
        if (!vote) break;
        const int k = __shfl(first, __ffsll((long long)vote) - 1, 64);
        long long q[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
#pragma unroll
        for (int r = 0; r < VROWS; ++r) {
            if ((pending & (1u << r)) && best_k[r] == k) {
                const long long t = (long long)trunc(pv[r] * fscale);     // two limbs of the global sum: t = hi * 2^24 + lo
                q[0] += 1; q[1] += z; q[2] += y0 + r; q[3] += x; q[4] += t >> 24; q[5] += t & 0xffffff;
                pending &= ~(1u << r);
            }
        }
        long long tot = wave_reduce8_i64(q);
        if ((lane & 7) == 0 && (lane >> 3) < 6 && tot != 0) atomic_add_i64(s.acc + (size_t)k * 6 + (lane >> 3), tot);
    }
}

int launch_vol_slic(VolState s, const double *vol, int32_t *labels, int max_iter, hipStream_t st)
{
    size_t n = (size_t)s.D * s.H * s.W;
    HIP_TRY(hipMemsetAsync(labels, 0xff, n * sizeof(int32_t), st));
    hipLaunchKernelGGL(k_vol_centroid_init, cdiv(s.K, 256), 256, 0, st, s);
    dim3 grid(cdiv(s.W, 64), cdiv(s.H, 4 * VROWS) * s.D);
    const size_t n_bricks = (size_t)s.nbz * s.nby * s.nbx;
    for (int it = 0; it < max_iter; ++it) {
        HIP_TRY(hipMemsetAsync(s.brick_count, 0, n_bricks * sizeof(int), st));
        hipLaunchKernelGGL(k_vol_scatter, cdiv(s.K, 256), 256, 0, st, s);
        if (it + 1 < max_iter) {
            hipLaunchKernelGGL(k_vol_assign<true>, grid, 256, 0, st, s, vol, labels);
            hipLaunchKernelGGL(k_vol_centroid_finalize, cdiv(s.K, 256), 256, 0, st, s);
        } else {
            hipLaunchKernelGGL(k_vol_assign<false>, grid, 256, 0, st, s, vol, labels);
        }
    }
    HIP_TRY(hipGetLastError());
    return 0;
}

// ---- float32 volumes: scikit-image 0.18 keeps a float32 input in float32 from end to end ---------------------
// (slic_superpixels.py `dtype = image.dtype` -> `_slic_cython[float32]`; oracle: orc_slic_gray3d_f32).  What that
// means for the arithmetic, all of it reproduced here:
//   * the Gaussian filter computes every line in double and stores float32 after every axis pass;
//   * `image * ratio` is a float32 product;
//   * centroids, spacing, the distance  ((dz + dy) + dx) * (1/step^2) + (v - cv)^2  and the search windows are float32;
//   * the centroid update adds the members of a segment in RASTER ORDER into float32 running sums -- not
//     associative, so no parallel reduction reproduces it.  Here one wave owns one centroid: it walks the
//     bounding box of the segment's voxels (tracked by the assignment kernel with integer atomics) in raster
//     order, 64 voxels per load, and folds the member lanes one by one (ballot bits -> v_readlane) into four
//     float32 accumulators kept in lanes 0..3 (z, y, x, value): the additions happen in exactly the order of the
//     sequential loop of _slic.pyx.
template <int AXIS>
__global__ void __launch_bounds__(256)
k_vol_blur_r32(const float *__restrict__ src, float *__restrict__ dst32, int D, int H, int W, Taps t, float fratio, int last)
{
    // (round 5: the values between two axis passes ARE float32 -- scipy stores every line in the output dtype -- so they are kept
    // as float32, not as doubles that hold float32 values: half the bytes per pass, and the float32 input is read as it is)
    // (grid: x blocks of a row, y, z -- no 64-bit division to find the voxel)
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y, z = blockIdx.z;
    if (x >= W) return;
    const size_t i = ((size_t)z * H + y) * W + x;
    double v;
    if (t.r < 0) {
        v = (double)src[i];                               // axis not filtered: the float32 value passes through
    } else {
        v = (double)src[i] * t.w[0];
        const int pos = AXIS == 0 ? z : AXIS == 1 ? y : x, len = AXIS == 0 ? D : AXIS == 1 ? H : W;
        if (pos - t.r >= 0 && pos + t.r < len) {
            // inside the volume along this axis (all but 2 r positions per line): the taps are i -/+ j * stride, no reflected
            // index, no 64-bit multiply per tap -- the same additions in the same order
            const size_t stride = AXIS == 0 ? (size_t)H * W : AXIS == 1 ? (size_t)W : 1;
            for (int j = t.r; j >= 1; --j) v += ((double)src[i - j * stride] + (double)src[i + j * stride]) * t.w[j];
        } else
        for (int j = t.r; j >= 1; --j) {
            size_t a, b;
            if (AXIS == 0) {
                a = ((size_t)vreflect(z - j, D) * H + y) * W + x;
                b = ((size_t)vreflect(z + j, D) * H + y) * W + x;
            } else if (AXIS == 1) {
                a = ((size_t)z * H + vreflect(y - j, H)) * W + x;
                b = ((size_t)z * H + vreflect(y + j, H)) * W + x;
            } else {
                a = ((size_t)z * H + y) * W + vreflect(x - j, W);
                b = ((size_t)z * H + y) * W + vreflect(x + j, W);
            }
            v += ((double)src[a] + (double)src[b]) * t.w[j];
        }
    }
    const float f = (float)v;                             // scipy stores the line in the output dtype (float32)
    dst32[i] = last ? f * fratio : f;                     // numpy: float32 array * Python float
}

// Round 6: the same three passes with a third of the traffic.  The axis passes above each read and write the whole volume (and the
// z pass fetches every plane nine times from beyond the L2: a plane of config 5 is 64 MB) -- 20.7 ms for 2^30 voxels.
//   * z pass (k_vol_blur_z32): a thread owns a column (four neighbouring columns: one 16-byte load per plane) and walks it along z
//     with the 2 R + 1 planes of its window in registers -- every voxel is read once;
//   * y and x pass together (k_vol_blur_yx32): a 64 x 32 tile of one slice with its halo goes through LDS, the y pass writes the
//     float32 lines scipy would store into a second LDS tile, the x pass reads them from there.
// Same operations in the same order on the same float32 intermediate values as the three passes (tests: both forms, bit for bit).
constexpr int VBLUR_R = 4;                         // largest radius the two kernels take (sigma = 1: radius 4)

template <int R, int VEC>
__global__ void __launch_bounds__(256)
k_vol_blur_z32(const float *__restrict__ src, float *__restrict__ dst, int D, int H, int W, Taps t, int zchunk)
{
    typedef float vec_t __attribute__((ext_vector_type(VEC)));
    const int x = (blockIdx.x * 256 + threadIdx.x) * VEC, y = blockIdx.y;
    const int z0 = blockIdx.z * zchunk, z1 = min(z0 + zchunk, D);
    if (x >= W) return;
    const size_t plane = (size_t)H * W, col = (size_t)y * W + x;
    auto load = [&](int z) { return *reinterpret_cast<const vec_t *>(src + (size_t)vreflect(z, D) * plane + col); };
    vec_t win[2 * R + 1];
#pragma unroll
    for (int j = 0; j <= 2 * R; ++j) win[j] = load(z0 - R + j);
    vec_t ahead = load(z0 + R + 1);                   // the plane the NEXT step needs, requested a step early
    for (int z = z0; z < z1; ++z) {
        vec_t out;
#pragma unroll
        for (int c = 0; c < VEC; ++c) {
            double v = (double)win[R][c] * t.w[0];
#pragma unroll
            for (int j = R; j >= 1; --j) v += ((double)win[R - j][c] + (double)win[R + j][c]) * t.w[j];
            out[c] = (float)v;                       // scipy stores the line in the output dtype (float32)
        }
        *reinterpret_cast<vec_t *>(dst + (size_t)z * plane + col) = out;
#pragma unroll
        for (int j = 0; j < 2 * R; ++j) win[j] = win[j + 1];
        win[2 * R] = ahead;
        if (z + 2 < z1) ahead = load(z + R + 2);
    }
}

// (radii known when compiled, a wave = eight rows of the tile: the y pass slides a window of 8 + 2 RY values down a column -- two LDS
// reads per value written where the loop over runtime radii took 2 RY + 1 --, no division for the coordinates: 6.2 -> see
// profiles/README_r06.md at 2^30 voxels.  Same sums in the same order.)
template <int RY, int RX>
__global__ void __launch_bounds__(256)
k_vol_blur_yx32(const float *__restrict__ src, float *__restrict__ dst, int H, int W, Taps ty, Taps tx, float fratio)
{
    constexpr int TW = 64 + 2 * RX, TH = 32 + 2 * RY;
    __shared__ float A[TH][TW];
    __shared__ float B[32][TW];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int X0 = blockIdx.x * 64, Y0 = blockIdx.y * 32;
    const float *__restrict__ sl = src + (size_t)blockIdx.z * H * W;
    const bool extra = lane < 2 * RX;                          // the lanes that also serve the columns 64 .. TW - 1
    const int xa = vreflect(X0 - RX + lane, W), xb = extra ? vreflect(X0 - RX + 64 + lane, W) : 0;
    // (a compile-time trip count: the rows of a wave are requested together and stored as they arrive -- `for (r = wave; r < TH; r += 4)`
    // was a load, a wait and a store per row, ten trips to memory one after the other)
    constexpr int TR = (TH + 3) / 4;
    float a0[TR], a1[TR];
#pragma unroll
    for (int i = 0; i < TR; ++i) {
        const int r = wave + 4 * i;
        const float *__restrict__ row = sl + (size_t)vreflect(Y0 - RY + min(r, TH - 1), H) * W;
        a0[i] = row[xa];
        a1[i] = extra ? row[xb] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < TR; ++i) {
        const int r = wave + 4 * i;
        if (r < TH) {
            A[r][lane] = a0[i];
            if (extra) A[r][64 + lane] = a1[i];
        }
    }
    __syncthreads();
#pragma unroll
    for (int part = 0; part < 2; ++part) {
        if (part == 1 && !extra) break;
        const int c = part ? 64 + lane : lane;
        float win[8 + 2 * RY];
#pragma unroll
        for (int i = 0; i < 8 + 2 * RY; ++i) win[i] = A[8 * wave + i][c];
#pragma unroll
        for (int o = 0; o < 8; ++o) {
            double v = (double)win[o + RY] * ty.w[0];
#pragma unroll
            for (int j = RY; j >= 1; --j) v += ((double)win[o + RY - j] + (double)win[o + RY + j]) * ty.w[j];
            B[8 * wave + o][c] = (float)v;                     // scipy stores the line in the output dtype (float32)
        }
    }
    __syncthreads();
    float *__restrict__ out = dst + (size_t)blockIdx.z * H * W;
    if (X0 + lane >= W) return;
#pragma unroll
    for (int o = 0; o < 8; ++o) {
        const int yy = 8 * wave + o;
        if (Y0 + yy >= H) break;
        double v = (double)B[yy][lane + RX] * tx.w[0];
#pragma unroll
        for (int j = RX; j >= 1; --j) v += ((double)B[yy][lane + RX - j] + (double)B[yy][lane + RX + j]) * tx.w[j];
        out[(size_t)(Y0 + yy) * W + X0 + lane] = (float)v * fratio;      // numpy: float32 array * Python float
    }
}

template <int RY>
static void launch_blur_yx32(int rx, dim3 grid, hipStream_t st, const float *src, float *dst, int H, int W, const Taps &ty, const Taps &tx,
                             float fratio)
{
    switch (rx) {
    case 0: hipLaunchKernelGGL((k_vol_blur_yx32<RY, 0>), grid, 256, 0, st, src, dst, H, W, ty, tx, fratio); break;
    case 1: hipLaunchKernelGGL((k_vol_blur_yx32<RY, 1>), grid, 256, 0, st, src, dst, H, W, ty, tx, fratio); break;
    case 2: hipLaunchKernelGGL((k_vol_blur_yx32<RY, 2>), grid, 256, 0, st, src, dst, H, W, ty, tx, fratio); break;
    case 3: hipLaunchKernelGGL((k_vol_blur_yx32<RY, 3>), grid, 256, 0, st, src, dst, H, W, ty, tx, fratio); break;
    default: hipLaunchKernelGGL((k_vol_blur_yx32<RY, 4>), grid, 256, 0, st, src, dst, H, W, ty, tx, fratio); break;
    }
}

template <int VEC>
static void launch_blur_z32(int R, dim3 grid, hipStream_t st, const float *src, float *dst, int D, int H, int W, const Taps &t, int zchunk)
{
    switch (R) {
    case 0: hipLaunchKernelGGL((k_vol_blur_z32<0, VEC>), grid, 256, 0, st, src, dst, D, H, W, t, zchunk); break;
    case 1: hipLaunchKernelGGL((k_vol_blur_z32<1, VEC>), grid, 256, 0, st, src, dst, D, H, W, t, zchunk); break;
    case 2: hipLaunchKernelGGL((k_vol_blur_z32<2, VEC>), grid, 256, 0, st, src, dst, D, H, W, t, zchunk); break;
    case 3: hipLaunchKernelGGL((k_vol_blur_z32<3, VEC>), grid, 256, 0, st, src, dst, D, H, W, t, zchunk); break;
    default: hipLaunchKernelGGL((k_vol_blur_z32<4, VEC>), grid, 256, 0, st, src, dst, D, H, W, t, zchunk); break;
    }
}

int launch_vol_preprocess_f32(const float *src, int D, int H, int W, const Taps &tz, const Taps &ty, const Taps &tx, double ratio,
                              double *bufA, double *bufB, hipStream_t st)
{
    float *a32 = reinterpret_cast<float *>(bufA), *b32 = reinterpret_cast<float *>(bufB);
    if (H > 65535 || D > 65535) {
        set_error("volume pre-processing: more than 65 535 rows or slices");
        return -1;
    }
    if (tz.r <= VBLUR_R && ty.r <= VBLUR_R && tx.r <= VBLUR_R && !knobs().pre_3pass) {
        // an axis that is not filtered (radius < 0) passes its values through: the single tap 1.0 (x * 1.0 == x, (float)(double)x == x)
        auto eff = [](const Taps &t) {
            Taps e = t;
            if (e.r < 0) {
                e.r = 0;
                e.w[0] = 1.0;
            }
            return e;
        };
        const float *mid = src;
        if (tz.r >= 0) {
            const int vec = (W % 4 == 0 && (reinterpret_cast<uintptr_t>(src) & 15) == 0) ? 4 : 1;
            // enough columns to fill the device, else the z range in chunks (each re-reads the 2 R planes around it)
            const long columns = (long)cdiv(W, vec) * H;
            int chunks = (int)std::min<long>(std::max<long>(1, (long)(16384L * 64 / std::max<long>(columns, 1))), std::max(1, D / 16));
            const int zchunk = cdiv(D, chunks);
            const dim3 grid(cdiv(cdiv(W, vec), 256), H, cdiv(D, zchunk));
            if (vec == 4) launch_blur_z32<4>(tz.r, grid, st, src, a32, D, H, W, tz, zchunk);
            else launch_blur_z32<1>(tz.r, grid, st, src, a32, D, H, W, tz, zchunk);
            mid = a32;
        }
        const dim3 tiles(cdiv(W, 64), cdiv(H, 32), D);
        const Taps ey = eff(ty), ex = eff(tx);
        switch (ey.r) {
        case 0: launch_blur_yx32<0>(ex.r, tiles, st, mid, b32, H, W, ey, ex, (float)ratio); break;
        case 1: launch_blur_yx32<1>(ex.r, tiles, st, mid, b32, H, W, ey, ex, (float)ratio); break;
        case 2: launch_blur_yx32<2>(ex.r, tiles, st, mid, b32, H, W, ey, ex, (float)ratio); break;
        case 3: launch_blur_yx32<3>(ex.r, tiles, st, mid, b32, H, W, ey, ex, (float)ratio); break;
        default: launch_blur_yx32<4>(ex.r, tiles, st, mid, b32, H, W, ey, ex, (float)ratio); break;
        }
        HIP_TRY(hipGetLastError());
        return 0;   // float32 result in bufB
    }
    const dim3 rows(cdiv(W, 256), H, D);
    hipLaunchKernelGGL(k_vol_blur_r32<0>, rows, 256, 0, st, src, b32, D, H, W, tz, 0.f, 0);
    hipLaunchKernelGGL(k_vol_blur_r32<1>, rows, 256, 0, st, (const float *)b32, a32, D, H, W, ty, 0.f, 0);
    hipLaunchKernelGGL(k_vol_blur_r32<2>, rows, 256, 0, st, (const float *)a32, b32, D, H, W, tx, (float)ratio, 1);
    HIP_TRY(hipGetLastError());
    return 0;   // float32 result in bufB
}

__device__ __forceinline__ void vol_window_f32(const VolState &s, float cz, float cy, float cx, int *w)
{
    // _slic.pyx with float32 centroids: <Py_ssize_t>max(cz - 2 * step_z, 0), <Py_ssize_t>min(cz + 2 * step_z + 1, depth)
    float a;
    a = cz - (float)(2 * s.step_z); w[0] = (int)(a > 0 ? a : 0.f);
    a = cz + (float)(2 * s.step_z); a = a + 1.f; w[1] = (int)(a < (float)s.D ? a : (float)s.D);
    a = cy - (float)(2 * s.step_y); w[2] = (int)(a > 0 ? a : 0.f);
    a = cy + (float)(2 * s.step_y); a = a + 1.f; w[3] = (int)(a < (float)s.H ? a : (float)s.H);
    a = cx - (float)(2 * s.step_x); w[4] = (int)(a > 0 ? a : 0.f);
    a = cx + (float)(2 * s.step_x); a = a + 1.f; w[5] = (int)(a < (float)s.W ? a : (float)s.W);
}

__device__ __forceinline__ void vol_bbox_reset(int *b)
{
    b[0] = b[2] = b[4] = 0x7fffffff;
    b[1] = b[3] = b[5] = -1;
}

__global__ void k_vol_centroid_init_f32(VolState s)
{
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= s.K) return;
    int ix = k % s.grid_n[2], iy = (k / s.grid_n[2]) % s.grid_n[1], iz = k / (s.grid_n[2] * s.grid_n[1]);
    float cz = (float)(double)(s.grid_0[0] + iz * s.grid_d[0]);
    float cy = (float)(double)(s.grid_0[1] + iy * s.grid_d[1]);
    float cx = (float)(double)(s.grid_0[2] + ix * s.grid_d[2]);
    s.cen32[(size_t)k * 4 + 0] = cz;
    s.cen32[(size_t)k * 4 + 1] = cy;
    s.cen32[(size_t)k * 4 + 2] = cx;
    s.cen32[(size_t)k * 4 + 3] = 0.f;
    vol_window_f32(s, cz, cy, cx, s.win + (size_t)k * 6);
    vol_bbox_reset(s.bbox + (size_t)k * 6);
}

// Round 3.  (1) The four waves of a workgroup (64 x 16 voxels of one slice = the cross-section of a brick) scan the brick's list
// ONCE, together, and stage the records of the centroids whose window meets the cross-section in LDS; before, every wave scanned
// the whole list (~230 windows of 24 bytes) for itself.  (2) Each wave then walks the staged candidates in ascending order of
// their lower bound over ITS strip (64 x 4) by repeated wave minima, and stops at the first bound above the worst best distance
// of the strip (`wave_worst`): every candidate behind it is at least as far.  ~150 windows meet a strip at sp_size 15, ~30 have a
// bound below the final worst distance; before, the list order decided how many were evaluated in full.  The pruning is exact as
// before: a candidate is skipped only when its bound EXCEEDS what every voxel of the strip already has (float32 operations, all
// monotone; ties go on being evaluated).  (3) The bounding boxes of the segments: the distinct labels of a strip are enumerated
// with ballots (a strip is 4 rows: row masks give the x and y extent), parked one per lane, and the lanes compare with / update
// the boxes in ONE round of loads instead of a serial load-compare-atomic chain per label.
struct VolRec {
    float cz, cy, cx, cv;
    int wy0, wy1, wx0, wx1;
};
constexpr int VLIST32 = 512;          // staged candidates per batch (a batch is walked when fewer than 256 slots are left)

// -DIMSEGM_VOL_PHASE_PROF (tools/build_variant.sh; never in the shipped library): shader-clock ticks per section of the assignment
// kernel, summed over the lane 0 of every wave, + waves and evaluated candidates -> imsegm_debug_vol_phases
#ifdef IMSEGM_VOL_PHASE_PROF
__device__ unsigned long long g_vol_phase[16];
#define VOL_PH_BEGIN unsigned long long ph_t0 = clock64(), ph_t1; int ph_evaluated = 0;
#define VOL_PH(i) do { if (lane == 0) { ph_t1 = clock64(); atomicAdd(&g_vol_phase[i], ph_t1 - ph_t0); ph_t0 = ph_t1; } } while (0)
#define VOL_PH_COUNT(i, v) do { if (lane == 0) atomicAdd(&g_vol_phase[i], (unsigned long long)(v)); } while (0)
#else
#define VOL_PH_BEGIN
#define VOL_PH(i)
#define VOL_PH_COUNT(i, v)
#endif

// Round 6: the walk over one batch of staged candidates, instantiated for the number of list slots a lane really holds (NS =
// ceil(count / 64): ~150 windows meet a cross-section at sp_size 15, i.e. 3 slots, where rounds 3 - 5 always carried the 8 of a full
// list through every minimum, search and clear), and on integer KEYS: the bits of the (non-negative) float32 bound with the slot
// number in their three lowest bits.  The wave minimum of the keys is the next candidate's bound -- rounded down by at most seven
// units in the last place, still a lower bound -- AND its slot: no search for the slot, no second exchange.  Which candidates are
// evaluated can only grow by the rounding; the result does not depend on it (the comparison carries the centroid index).
constexpr int VOL_KEY_INF = 0x7f800000;

__device__ __forceinline__ int wave_min_key(int key)
{
    // keys of real candidates lie in [0, VOL_KEY_INF): as a maximum of VOL_KEY_INF - key, a lane without a source contributes 0
    int x = VOL_KEY_INF - key;
    x = max(x, __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, false));
    x = max(x, __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, false));
    x = max(x, __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, false));
    x = max(x, __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, false));
    x = max(x, __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false));
    x = max(x, __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false));
    return VOL_KEY_INF - __builtin_amdgcn_readlane(x, 63);
}

// VZ slices of one workgroup (see k_vol_assign_f32): a lane holds VZ x VROWS voxels -- one column of 4 rows in VZ consecutive
// slices.  `cov[c]`: bit zi set when slice z0 + zi lies inside candidate c's search window.
// Measured (round 6, 64 x 4096 x 4096, one box each): VZ = 1 11.4 ms per accumulating sweep; VZ = 2 -- the scan of the brick's list,
// its barriers and the flush of the boxes serving twice the voxels, a candidate selected once for eight rows of a lane -- 12.1 ms:
// 102 registers per lane leave four waves per SIMD where 56 leave eight, and the kernel lives on the waves that compute while
// others wait.  The code takes either; the label maps are the same (tests with both).
#ifndef VOL_ASSIGN_SLICES
#define VOL_ASSIGN_SLICES 1
#endif
constexpr int VZ = VOL_ASSIGN_SLICES;

// The voxels of a wave inside the workgroup's 64 x 16 cross-section.  VOL_ASSIGN_TILE 16 (round 6): a 16 x 16 TILE -- lane = (lane & 15)
// in x, rows 4 (lane >> 4) .. + 3 in y; 64: a 64 x 4 strip (rounds 3 - 6) -- lane in x, four rows.  The walk evaluates a candidate for
// all 256 voxels of the wave; at config 5 a supervoxel is ~35 x 35 voxels of a slice, a strip met the windows of ~9 centroids whose
// bound lay below its worst distance, a tile meets ~half of that.  Same candidates per VOXEL, same arithmetic, same comparison: the
// label maps do not move (tests with both).
#ifndef VOL_ASSIGN_TILE
#define VOL_ASSIGN_TILE 16
#endif
// (32: a 32 x 8 tile -- lane & 31 in x, rows 4 (lane >> 5) .. + 3 -- whose rows are whole 128-byte lines of the label map: the 16 x 16
// tile writes half lines, 2.2 bytes to memory for every byte of the map by the counters.)
constexpr bool VT16 = VOL_ASSIGN_TILE != 64;              // (a tile, not a strip)
static_assert(VOL_ASSIGN_TILE == 16 || VOL_ASSIGN_TILE == 32 || VOL_ASSIGN_TILE == 64, "tile of a wave: 16 x 16, 32 x 8 or 64 x 4");
constexpr int VT_W = VOL_ASSIGN_TILE, VT_H = (64 / VOL_ASSIGN_TILE) * VROWS;
constexpr int VT_SHIFT = VOL_ASSIGN_TILE == 16 ? 4 : VOL_ASSIGN_TILE == 32 ? 5 : 6;
// the neighbour to the left inside the wave's row of voxels (the first lane of a tile's row keeps `first`)
__device__ __forceinline__ int tile_prev(int v, int first)
{
    if (VOL_ASSIGN_TILE == 16) return __builtin_amdgcn_update_dpp(first, v, 0x111, 0xf, 0xf, false);      // row_shr:1
    const int p = lane_prev(v, first);
    return (VOL_ASSIGN_TILE == 32 && (threadIdx.x & 31) == 0) ? first : p;
}

template <int NS>
__device__ __forceinline__ void vol_walk_f32(const VolRec *rec, const int *list, const unsigned char *cov, int count, int lane, int y0,
                                             int y1w, int x0w, int x1w, int yl, int x, bool xin, int H, int nzv, float fz0, float fx, float sz,
                                             float sy, float sx, float sw, const float (&pv)[VZ][VROWS], float (&best_d)[VZ][VROWS],
                                             int (&best_k)[VZ][VROWS], float &wave_worst)
{
    static_assert(NS >= 1 && NS <= 8, "three bits of a key hold the slot");
    // bounds over this wave's voxels [y0, y1w) x [x0w, x1w) (per slice); a window that misses them is out.  The bound of a candidate
    // is its smallest bound over the slices its window covers: below every distance it can give a voxel of this wave.
    // (yl: the first of the lane's own VROWS rows)
    int key[NS];
#pragma unroll
    for (int j = 0; j < NS; ++j) {
        const int c = lane + 64 * j;
        key[j] = VOL_KEY_INF;
        if (c < count) {
            const VolRec rc = rec[c];
            const int cv = VZ > 1 ? cov[c] : 1;            // (one slice: the scan has kept the windows that cover it)
            if (rc.wy0 < y1w && rc.wy1 > y0 && rc.wx0 < x1w && rc.wx1 > x0w) {
                const float yn = fminf(fmaxf(rc.cy, (float)y0), (float)(y1w - 1));
                const float xn = fminf(fmaxf(rc.cx, (float)x0w), (float)(x1w - 1));
                const float tyl = sy * (rc.cy - yn), txl = sx * (rc.cx - xn);
                const float dy2 = tyl * tyl, dx2 = txl * txl;
                float lb = INFINITY;
#pragma unroll
                for (int zi = 0; zi < VZ; ++zi) {
                    const float tz = sz * (rc.cz - (fz0 + (float)zi));
                    const float dz = tz * tz;
                    const float one = ((dz + dy2) + dx2) * sw;          // >= 0, or +inf / NaN (then: not a candidate)
                    lb = ((cv >> zi) & 1) ? fminf(lb, one) : lb;
                }
                const int bits = __float_as_int(lb);
                key[j] = (bits >= 0 && bits < VOL_KEY_INF) ? ((bits & ~7) | j) : VOL_KEY_INF;
            }
        }
    }
    int since_refresh = 0;
    while (true) {
        int mine = key[0];
#pragma unroll
        for (int j = 1; j < NS; ++j) mine = min(mine, key[j]);
        const int wk = wave_min_key(mine);
        if (wk >= VOL_KEY_INF) break;                              // nothing left in the batch
        if (!(__int_as_float(wk & ~7) <= wave_worst)) break;       // every candidate left is farther than what the strips have
        const unsigned long long own = __ballot(mine == wk);
        const int src = __ffsll((long long)own) - 1;
        const int jsel = wk & 7;
#pragma unroll
        for (int j = 0; j < NS; ++j)
            if (lane == src && j == jsel) key[j] = VOL_KEY_INF;
        const int c = src + 64 * jsel;
        const int ck = list[c];
        const VolRec rc = rec[c];
        const int cv = VZ > 1 ? cov[c] : 1;
        VOL_PH_COUNT(11, 1);
        const bool inx = x >= rc.wx0 && x < rc.wx1;
        const float tx = sx * (rc.cx - fx);
        const float dx2 = tx * tx;
        float dy[VROWS];
#pragma unroll
        for (int r = 0; r < VROWS; ++r) {
            const float ty = sy * (rc.cy - (float)(yl + r));
            dy[r] = ty * ty;
        }
#pragma unroll
        for (int zi = 0; zi < VZ; ++zi) {
            if (!((cv >> zi) & 1)) continue;                       // (wave uniform: the slice is outside the window)
            const float tz = sz * (rc.cz - (fz0 + (float)zi));
            const float dz = tz * tz;
#pragma unroll
            for (int r = 0; r < VROWS; ++r) {
                const int y = yl + r;
                const bool iny = y >= rc.wy0 && y < rc.wy1;
                if (!VT16 && !iny) continue;                         // (a strip's row: wave uniform)
                float d = ((dz + dy[r]) + dx2) * sw;
                const float t = pv[zi][r] - rc.cv;
                d = d + t * t;
                const bool take = inx && iny && (best_d[zi][r] > d || (best_d[zi][r] == d && ck < best_k[zi][r]));
                // (the compiler guards the arithmetic of a row with the window test -- branches; written with & and | instead, as
                // compares into masks and two selects, every row is computed for every candidate and the kernel spills: 14.7
                // against 8.7 ms per sweep at config 5)
                best_d[zi][r] = take ? d : best_d[zi][r];
                best_k[zi][r] = take ? ck : best_k[zi][r];
            }
        }
        if (++since_refresh == 2) {
            since_refresh = 0;
            float m2 = 0.f;
#pragma unroll
            for (int zi = 0; zi < VZ; ++zi)
#pragma unroll
                for (int r = 0; r < VROWS; ++r)
                    if (xin && (yl + r) < H && zi < nzv) m2 = fmaxf(m2, best_d[zi][r]);
            wave_worst = wave_max_nonneg_f32(m2);           // (distances: never negative; +inf while a voxel has no candidate)
        }
    }
}


// Round 6.  A phase profile of the kernel (shader-clock ticks per section, tools/vol_phase_probe.py) showed where a workgroup's
// 25 us go: 27 % into the scan of the brick list (list entry -> search window of that centroid -> test: two dependent trips to
// global memory), 8 % into fetching the records of the hits (a third), 31 % into the bounding boxes (per strip row, every run
// start loads its label's box and compares: four more dependent trips per wave) and 17 % into the walk itself -- the kernel waits,
// all four waves of a workgroup together, far more than it computes.  So the trips are taken out:
//   * k_vol_scatter_f32 writes everything the assignment needs of a centroid -- position, value, search window, index: one
//     VolEntry of 48 bytes -- into the brick lists; the scan loads an entry with three 16-byte loads, tests it and drops it
//     straight into the LDS records: ONE trip instead of three, and one barrier less;
//   * the bounding boxes of a workgroup's labels meet in an LDS hash table first (a 64 x 16 cross-section sees a dozen labels):
//     run starts update it with LDS atomics, and one thread per label compares with / updates the global box at the end --
//     a dozen box updates per workgroup instead of one per run, row and wave (~100).
// Same candidates, same arithmetic, same order-free comparison: the label maps do not move.
struct VolEntry {
    float cz, cy, cx, cv;
    int wy0, wy1, wx0, wx1;
    int wz0, wz1, k, pad;
};
static_assert(sizeof(VolEntry) == 48, "three 16-byte loads per entry");
constexpr int VT_SLOTS = 64;          // labels of a workgroup's cross-section whose boxes meet in LDS (more: straight to global memory)

// every centroid writes its entry into the list of each brick its search window meets (float32 volumes).  One WAVE per centroid:
// lane j takes the bricks j, j + 64, ... of the window's brick range (~50 at sp_size 15: a thread per centroid walked them one
// after the other, an atomic and three 16-byte stores each -- 0.98 ms per sweep at 298 116 supervoxels)
__global__ void __launch_bounds__(256) k_vol_scatter_f32(VolState s)
{
    const int k = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (k >= s.K) return;
    const int *w = s.win + (size_t)k * 6;
    const int w0 = w[0], w1 = w[1], w2 = w[2], w3 = w[3], w4 = w[4], w5 = w[5];
    if (w1 <= w0 || w3 <= w2 || w5 <= w4) return;       // dead centroid: empty window
    const float4 cen = *reinterpret_cast<const float4 *>(s.cen32 + (size_t)k * 4);
    const int4 e0 = make_int4(__float_as_int(cen.x), __float_as_int(cen.y), __float_as_int(cen.z), __float_as_int(cen.w));
    const int4 e1 = make_int4(w2, w3, w4, w5);
    const int4 e2 = make_int4(w0, w1, k, 0);
    const int bz0 = w0 / VOL_BZ, by0 = w2 / VOL_BY, bx0 = w4 / VOL_BX;
    const int nz = (w1 - 1) / VOL_BZ - bz0 + 1, ny = (w3 - 1) / VOL_BY - by0 + 1, nx = (w5 - 1) / VOL_BX - bx0 + 1;
    const int total = nz * ny * nx;
    for (int j = lane; j < total; j += 64) {
        const int bx = bx0 + j % nx, by = by0 + (j / nx) % ny, bz = bz0 + j / (nx * ny);
        const int b = (bz * s.nby + by) * s.nbx + bx;
        const int pos = atomicAdd(&s.brick_count[b], 1);
        if (pos < s.brick_cap) {
            int4 *dst = reinterpret_cast<int4 *>(s.brick_entries + ((size_t)b * s.brick_cap + pos) * 12);
            dst[0] = e0;
            dst[1] = e1;
            dst[2] = e2;
        }
    }
}

// waves per SIMD the register allocation aims at.  On 64 x 4 strips: 80 registers when left alone (six waves), 72 at seven -- 2.53
// against 2.68 ms per sweep of a quarter volume, alternating on one box --, spills at eight (64 registers: 3.47 ms).  On 16 x 16
// tiles with the labels staged through LDS (more per-lane state: own rows, tile geometry) seven waves spill into scratch inside the
// walk: 6 -> 7.48, 7 -> 8.72, 5 -> 8.37, 8 -> 11.97 ms per sweep at config 5, one box each pair.
#ifndef VOL_ASSIGN_WAVES
#define VOL_ASSIGN_WAVES 6
#endif
#define VOL_ASSIGN_ATTR __attribute__((amdgpu_waves_per_eu(VOL_ASSIGN_WAVES, VOL_ASSIGN_WAVES)))
template <bool TRACK>
__global__ void __launch_bounds__(256) VOL_ASSIGN_ATTR
k_vol_assign_f32(VolState s, const float *__restrict__ vol, int32_t *__restrict__ labels)
{
    // A workgroup = the 64 x 16 cross-section of a brick in VZ consecutive slices (VZ = 1; 2 was measured, see VZ above);
    // a wave = 64 columns x 4 rows x VZ slices.
    __shared__ int list[VLIST32];
    __shared__ __attribute__((aligned(16))) VolRec rec[VLIST32];      // (filled with 16-byte stores)
    // (one slice: no coverage bytes and no z columns in the table of the boxes -- 19.7 KB per workgroup, eight of them per CU; with
    // them it is 20.8 KB and seven)
    __shared__ unsigned char cov[VZ > 1 ? VLIST32 : 1];
    __shared__ int wave_base[4];
    constexpr int HB = VZ > 1 ? 6 : 4, HB_Y = VZ > 1 ? 2 : 0;                        // label -> [zmin, zmax,] ymin, ymax, xmin, xmax
    // (tiles of one slice: the box of a label inside the cross-section as the SET of its columns -- 64 bits, hb_box[][0..1] -- and of
    // its rows -- 16 bits, hb_box[][2] --, joined with OR)
    constexpr bool HB_SETS = VT16 && VZ == 1;
    __shared__ int hb_key[TRACK ? VT_SLOTS : 1];
    __shared__ __attribute__((aligned(8))) int hb_box[TRACK ? VT_SLOTS : 1][HB];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    VOL_PH_BEGIN
    const int rows_per_block = 4 * VROWS;
    const int yb = cdiv(s.H, rows_per_block);
    const int z0 = (blockIdx.y / yb) * VZ;
    const int nzv = min(VZ, s.D - z0);                        // slices of this workgroup inside the volume
    const int Y0 = (blockIdx.y % yb) * rows_per_block, Y1 = min(Y0 + rows_per_block, s.H);
    // the wave's voxels: [y0, y1w) x [xt0, xt1); the lane's: column x, rows yl .. yl + VROWS - 1
    constexpr int TILES_X = 64 / VT_W;                        // tiles of a workgroup side by side (4, 2 or 1), 4 / TILES_X below each other
    const int y0 = Y0 + (wave / TILES_X) * VT_H;
    const int yl = y0 + (lane >> VT_SHIFT) * VROWS;
    const int x0w = blockIdx.x * 64, x1w = min(x0w + 64, s.W);                    // the cross-section (the scan of the brick's list)
    const int xt0 = x0w + (wave % TILES_X) * VT_W, xt1 = min(xt0 + VT_W, s.W);
    const int x = xt0 + (lane & (VT_W - 1));
    const bool alive = xt0 < s.W && y0 < s.H;                 // (a wave outside the volume takes part in the barriers only)
    const int y1w = min(y0 + VT_H, s.H);
    const bool xin = x < s.W;
    if (TRACK && tid < VT_SLOTS) {                            // (visible to all after the first barrier of the scan below)
        hb_key[tid] = -1;
#pragma unroll
        for (int j = 0; j < HB; ++j) hb_box[tid][j] = HB_SETS ? 0 : (j & 1) ? -1 : 0x7fffffff;
    }
    float pv[VZ][VROWS], best_d[VZ][VROWS];
    int best_k[VZ][VROWS];
#pragma unroll
    for (int zi = 0; zi < VZ; ++zi)
#pragma unroll
        for (int r = 0; r < VROWS; ++r) {
            const bool ok = alive && xin && (yl + r) < s.H && zi < nzv;
            pv[zi][r] = vol[ok ? ((size_t)(z0 + zi) * s.H + yl + r) * s.W + x : 0];
            best_d[zi][r] = INFINITY;
            best_k[zi][r] = -1;
        }
    const float fz0 = (float)z0, fx = (float)x;
    const float sz = (float)s.sz, sy = (float)s.sy, sx = (float)s.sx;
    const float sw = (float)s.spatial_weight;              // = (float)(1 / ((double)step * (double)step))
    int count = 0;                                         // (uniform over the workgroup)
    float wave_worst = INFINITY;
    static_assert(VOL_BZ % VZ == 0, "the slices of a workgroup lie in one brick");
    const int brick = ((z0 / VOL_BZ) * s.nby + (Y0 / VOL_BY)) * s.nbx + blockIdx.x;
    const int bcount = s.brick_count[brick];
    const bool whole = bcount > s.brick_cap;                   // list overflow: scan every centroid
    const int4 *__restrict__ entries = reinterpret_cast<const int4 *>(s.brick_entries + (size_t)brick * s.brick_cap * 12);
    // (requesting the first 256 entries before the length of the list is known -- one trip to memory instead of two -- was built and
    // measured: 3.30 against 2.67 ms per sweep of a 64 x 2048 x 2048 volume on one box, alternating.  Twelve more registers per
    // thread and half the speculative loads wasted cost more than the trip.)
    const int nscan = whole ? s.K : bcount;
    const int nblk = cdiv(nscan, 256);
    static_assert(VLIST32 == 512, "eight list slots per lane: three bits of a key");
    VOL_PH(0);
    VOL_PH_COUNT(8, 1);
    VOL_PH_COUNT(9, nscan);
    if (nblk == 0) __syncthreads();                            // (the table of the boxes is initialised for everybody)
    for (int b = 0; b < nblk; ++b) {
        const int i = b * 256 + tid;
        int4 e0 = make_int4(0, 0, 0, 0), e1 = e0, e2 = e0;
        bool hit = false;
        if (i < nscan) {
            if (whole) {
                const int *w = s.win + (size_t)i * 6;
                const float4 cen = *reinterpret_cast<const float4 *>(s.cen32 + (size_t)i * 4);
                e0 = make_int4(__float_as_int(cen.x), __float_as_int(cen.y), __float_as_int(cen.z), __float_as_int(cen.w));
                e1 = make_int4(w[2], w[3], w[4], w[5]);
                e2 = make_int4(w[0], w[1], i, 0);
            } else {
                e0 = entries[3 * i];
                e1 = entries[3 * i + 1];
                e2 = entries[3 * i + 2];
            }
            hit = e2.x < z0 + nzv && e2.y > z0 && e1.x < Y1 && e1.y > Y0 && e1.z < x1w && e1.w > x0w;
        }
        const unsigned long long m = __ballot(hit);
        if (lane == 0) wave_base[wave] = __popcll(m);
        __syncthreads();
        int base = count;
        for (int w2 = 0; w2 < wave; ++w2) base += wave_base[w2];
        const int added = wave_base[0] + wave_base[1] + wave_base[2] + wave_base[3];
        if (hit) {
            const int pos = base + __popcll(m & ((1ULL << lane) - 1ULL));
            list[pos] = e2.z;
            int4 *dst = reinterpret_cast<int4 *>(&rec[pos]);
            dst[0] = e0;
            dst[1] = e1;
            int inside = 0;
#pragma unroll
            for (int zi = 0; zi < VZ; ++zi) inside |= (zi < nzv && z0 + zi >= e2.x && z0 + zi < e2.y) ? (1 << zi) : 0;
            if (VZ > 1) cov[pos] = (unsigned char)inside;
        }
        count += added;
        __syncthreads();                                     // (list and records are complete; wave_base may be rewritten)
        VOL_PH(1);
        if (count <= VLIST32 - 256 && b + 1 < nblk) continue;
        VOL_PH_COUNT(10, count);
        VOL_PH(2);
        if (alive) {
#define VOL_WALK(NS) vol_walk_f32<NS>(rec, list, cov, count, lane, y0, y1w, xt0, xt1, yl, x, xin, s.H, nzv, fz0, fx, sz, sy, sx, sw, pv, best_d, best_k, wave_worst)
            switch ((count + 63) >> 6) {
            case 0: break;
            case 1: VOL_WALK(1); break;
            case 2: VOL_WALK(2); break;
            case 3: VOL_WALK(3); break;
            case 4: VOL_WALK(4); break;
            case 5: VOL_WALK(5); break;
            case 6: VOL_WALK(6); break;
            case 7: VOL_WALK(7); break;
            default: VOL_WALK(8); break;
            }
#undef VOL_WALK
        }
        VOL_PH(3);
        count = 0;
        if (b + 1 < nblk) __syncthreads();                   // (everybody is done with this batch's list and records)
        VOL_PH(4);
    }
    unsigned pending = 0;                                    // bit zi * VROWS + r: the voxel carries a label
    // Tiles narrower than a line of the label map (16 voxels = 64 of its 128 bytes) hand their new labels to the LDS the records
    // have left, and every wave stores four whole rows of the cross-section: written half line by half line the map cost 2.3 bytes
    // to memory per byte (WRITE_SIZE, profiles/pmc_r06_cfg5_kernels.txt).
#ifndef VOL_ASSIGN_DIRECT_STORES
    constexpr bool STAGED = VOL_ASSIGN_TILE == 16 && VZ == 1;
#else
    constexpr bool STAGED = false;
#endif
    if (STAGED) {
        int *stage = reinterpret_cast<int *>(rec);           // [16][64]
        static_assert(sizeof(rec) >= 16 * 64 * sizeof(int), "the records' LDS holds a cross-section of labels");
        __syncthreads();                                     // (everybody is done with the last batch's records)
#pragma unroll
        for (int r = 0; r < VROWS; ++r) stage[(yl - Y0 + r) * 64 + (x - x0w)] = best_k[0][r];
        __syncthreads();
#pragma unroll
        for (int r = 0; r < VROWS; ++r) {
            const int ys = Y0 + wave * VROWS + r, xs = x0w + lane;
            const int k = stage[(wave * VROWS + r) * 64 + lane];
            if (k >= 0 && xs < s.W && ys < s.H) labels[((size_t)z0 * s.H + ys) * s.W + xs] = k;
        }
    }
    if (alive) {
#pragma unroll
        for (int zi = 0; zi < VZ; ++zi)
#pragma unroll
            for (int r = 0; r < VROWS; ++r) {
                if (!(xin && (yl + r) < s.H && zi < nzv)) continue;
                const size_t p = ((size_t)(z0 + zi) * s.H + yl + r) * s.W + x;
                if (best_k[zi][r] >= 0) {
                    if (!STAGED) labels[p] = best_k[zi][r];
                } else {
                    best_k[zi][r] = labels[p];                    // uncovered voxel keeps its previous assignment
                }
                if (best_k[zi][r] >= 0) pending |= 1u << (zi * VROWS + r);
            }
    }
    VOL_PH(5);
    if (!TRACK) return;
    // bounding box of every segment's voxels (incl. the ones that kept an old label): the region the order-preserving update of
    // that centroid has to walk.  By RUNS (round 5): in every row of the strip the first lane of a run of equal labels finds the
    // end of its run in the vote of the run starts.  Round 6: it updates the box of its label in the workgroup's LDS table; the
    // table goes to the global boxes once, below.  Minima and maxima: the boxes are the same whatever the order and however often
    // a label is met.
    // Tiles (one slice): by runs as the strips below, but a run joins the label's SETS of columns and rows with two ORs that return
    // nothing, and a lane that has found the slot of a label keeps it for the rows below (a label change costs the search again).
    // (Measured at config 5, one box: runs with minima / maxima as below 9.49 ms per sweep; label by label -- the first waiting lane
    // names a label, four votes give its columns and rows, one lane joins them -- 11.05 ms: a serial chain per label and wave.)
    if (HB_SETS && alive) {
        int slot_of = -2, slot = -1;
#pragma unroll
        for (int r = 0; r < VROWS; ++r) {
            const int k = (pending >> r) & 1u ? best_k[0][r] : -1;
            const int kp = tile_prev(k, -2);                        // (the first lane of a row of the tile: a start)
            const bool start = k >= 0 && kp != k;
            const unsigned long long starts = __ballot(start), valid = __ballot(k >= 0);
            if (start) {
                const unsigned long long stop = (starts | ~valid) & ~((2ULL << lane) - 1ULL);     // the lanes above this one
                const int len = (stop ? __ffsll((long long)stop) - 1 : 64) - lane;                 // (<= 16: the next row starts)
                if (k != slot_of) {
                    slot_of = k;
                    slot = (int)(((unsigned int)k * 2654435761u) >> 26);          // 6 bits
                    bool placed = false;
                    for (int probe = 0; probe < VT_SLOTS; ++probe) {
                        const int old = atomicCAS(&hb_key[slot], -1, k);
                        if (old == -1 || old == k) {
                            placed = true;
                            break;
                        }
                        slot = (slot + 1) & (VT_SLOTS - 1);
                    }
                    slot = placed ? slot : -1;
                }
                const int y = yl + r;
                if (slot >= 0) {
                    atomicOr(reinterpret_cast<unsigned long long *>(&hb_box[slot][0]), ((1ULL << len) - 1ULL) << (x - x0w));
                    atomicOr(reinterpret_cast<unsigned int *>(&hb_box[slot][2]), 1u << (y - Y0));
                } else {                                              // (more than VT_SLOTS labels in a cross-section)
                    int *bb = s.bbox + (size_t)k * 6;
                    atomicMin(&bb[0], z0); atomicMax(&bb[1], z0);
                    atomicMin(&bb[2], y); atomicMax(&bb[3], y);
                    atomicMin(&bb[4], x); atomicMax(&bb[5], x + len - 1);
                }
            }
        }
    }
    if (!HB_SETS && alive) {
#pragma unroll
        for (int zi = 0; zi < VZ; ++zi)
#pragma unroll
            for (int r = 0; r < VROWS; ++r) {
                const int k = (pending >> (zi * VROWS + r)) & 1u ? best_k[zi][r] : -1;
                const int kp = tile_prev(k, -2);                    // (the first lane of a row of the tile: a start)
                const bool start = k >= 0 && kp != k;
                const unsigned long long starts = __ballot(start), valid = __ballot(k >= 0);
                if (start) {
                    const unsigned long long stop = (starts | ~valid) & ~((2ULL << lane) - 1ULL);     // the lanes above this one
                    const int end = stop ? __ffsll((long long)stop) - 1 : 64;
                    const int z = z0 + zi, y = yl + r, xlo = x, xhi = x + (end - lane) - 1;
                    int slot = (int)(((unsigned int)k * 2654435761u) >> 26);          // 6 bits
                    bool placed = false;
                    for (int probe = 0; probe < VT_SLOTS; ++probe) {
                        const int old = atomicCAS(&hb_key[slot], -1, k);
                        if (old == -1 || old == k) {
                            placed = true;
                            break;
                        }
                        slot = (slot + 1) & (VT_SLOTS - 1);
                    }
                    if (placed) {
                        if (VZ > 1) {
                            atomicMin(&hb_box[slot][0], z);
                            atomicMax(&hb_box[slot][1], z);
                        }
                        atomicMin(&hb_box[slot][HB_Y], y);
                        atomicMax(&hb_box[slot][HB_Y + 1], y);
                        atomicMin(&hb_box[slot][HB_Y + 2], xlo);
                        atomicMax(&hb_box[slot][HB_Y + 3], xhi);
                    } else {                                          // (more than VT_SLOTS labels in a cross-section)
                        int *bb = s.bbox + (size_t)k * 6;
                        atomicMin(&bb[0], z); atomicMax(&bb[1], z);
                        atomicMin(&bb[2], y); atomicMax(&bb[3], y);
                        atomicMin(&bb[4], xlo); atomicMax(&bb[5], xhi);
                    }
                }
            }
    }
    __syncthreads();
    if (tid < VT_SLOTS && hb_key[tid] >= 0) {
        int *bb = s.bbox + (size_t)hb_key[tid] * 6;
        const int2 bz = *reinterpret_cast<const int2 *>(bb), by = *reinterpret_cast<const int2 *>(bb + 2),
                   bx = *reinterpret_cast<const int2 *>(bb + 4);
        const int zlo = VZ > 1 ? hb_box[tid][0] : z0, zhi = VZ > 1 ? hb_box[tid][1] : z0;
        int ylo = hb_box[tid][HB_Y], yhi = hb_box[tid][HB_Y + 1], xlo = hb_box[tid][HB_Y + 2], xhi = hb_box[tid][HB_Y + 3];
        if (HB_SETS) {
            const unsigned long long colset = *reinterpret_cast<const unsigned long long *>(&hb_box[tid][0]);
            const unsigned rows = (unsigned)hb_box[tid][2];
            ylo = Y0 + __ffs(rows) - 1;
            yhi = Y0 + 31 - __clz(rows);
            xlo = x0w + __ffsll((long long)colset) - 1;
            xhi = x0w + 63 - __clzll((long long)colset);
        }
        if (bz.x > zlo) atomicMin(&bb[0], zlo);
        if (bz.y < zhi) atomicMax(&bb[1], zhi);
        if (by.x > ylo) atomicMin(&bb[2], ylo);
        if (by.y < yhi) atomicMax(&bb[3], yhi);
        if (bx.x > xlo) atomicMin(&bb[4], xlo);
        if (bx.y < xhi) atomicMax(&bb[5], xhi);
    }
    VOL_PH(6);
}

// Round 5: one LANE per centroid.  The float32 sums of a segment must be formed in raster order (see above), which makes the sum of
// ONE segment a serial chain -- but the chains of different segments have nothing to do with each other.  Rounds 3 / 4 gave a
// whole wave to one chain: per member voxel a find-first-bit, a bit clear, two v_readlane and two packed additions, i.e. eight
// issue slots of which the wave used one lane's worth (13.2 ms per sweep at 298 116 supervoxels against 3.5 here; A/B in
// profiles/rocprof_r05_cfg5_kernel_stats.txt against rocprof_r04_cfg5_kernel_stats.txt; removed in round 6).  Here a lane walks the
// bounding box of ITS segment voxel by voxel -- labels[p] == k ? add : skip -- and the sixty-four chains of a wave advance together;
// lanes of one wave hold neighbouring centroids of a grid row, whose boxes lie side by side, so the lines a wave touches are shared
// by its lanes and reused by the next steps of the walk (L1 / L2), and the additions of a chain happen in exactly the order of the
// serial loop: z, then y, then x ascending.  Same results bit for bit (the parity tests of the float32 volumes; the full-size
// label map against scikit-image's).
constexpr int VU_STEP = 4;          // voxels of one 16-byte load
// four consecutive 4-byte elements at a 4-byte aligned address: one global_load_dwordx4
template <typename T> struct __attribute__((packed, aligned(4))) Quad {
    T v[4];
};
static_assert(VU_STEP == 4, "a round of the update is one Quad");
// (quads of a lane requested per round, config 5, ms per sweep: 1 -> 3.01, 2 -> 2.03, 4 -> 1.81, 6 -> 1.84, 8 -> 1.82)
// (the walk as one sequence of rounds with the NEXT round's labels requested before this round's values -- one trip per round
// instead of two -- measured 1.84 against 1.80: with four quads per round the trips are no longer what the kernel waits for)
#ifndef VOL_UPDATE_QUADS
#define VOL_UPDATE_QUADS 4
#endif
constexpr int VU_QUADS = VOL_UPDATE_QUADS;

// (lanes per workgroup, config 5, one box: 64 -> 3.26, 256 -> 2.93, 512 -> 3.13, 1 024 -> 3.89 ms per sweep; an eighth of the centroids
// per XCD -- block index modulo 8 -> a fixed range -- 3.06 against 2.99: not kept)
#ifndef VOL_UPDATE_BLOCK
#define VOL_UPDATE_BLOCK 256
#endif
__global__ void __launch_bounds__(VOL_UPDATE_BLOCK)
k_vol_update_f32_lane(VolState s, const float *__restrict__ vol, const int32_t *__restrict__ labels)
{
    const int k = blockIdx.x * VOL_UPDATE_BLOCK + threadIdx.x;
    if (k >= s.K) return;
    int *bb = s.bbox + (size_t)k * 6;
    int *w = s.win + (size_t)k * 6;
    const int z0 = bb[0], z1 = bb[1], y0 = bb[2], y1 = bb[3], x0 = bb[4], x1 = bb[5];
    if (z1 < z0) {                                            // no voxel carries this label: the centroid is dead
#pragma unroll
        for (int j = 0; j < 6; ++j) w[j] = 0;
        return;
    }
    float sz = 0.f, sy = 0.f, sx = 0.f, sv = 0.f;
    int cnt = 0;
    for (int z = z0; z <= z1; ++z) {
        const float fz = (float)z;
        for (int y = y0; y <= y1; ++y) {
            const float fy = (float)y;
            const size_t row = ((size_t)z * s.H + y) * s.W;
            // VU_STEP voxels per round: their labels and values are requested together (eight loads in flight instead of a chain
            // of two per voxel), then added in order.  A voxel of another segment adds +0.0f, which leaves a sum that started at
            // +0.0f bit for bit as it is (a sum of this kind is never -0.0f: (+0) + (-0) = +0).
            // (round 6: the four voxels of a round come in ONE 16-byte load of labels and one of values -- the lanes of a wave walk
            // different boxes, so every load instruction touches 64 cache lines whatever its width; a quarter of the instructions.
            // Voxels past x1 are loaded -- the buffers end in padding -- and masked.  The values of a quad are requested only where
            // one of its labels is the lane's -- under half of a box's voxels: 3.01 against 3.25 ms per sweep at config 5, one box.)
            // (VU_QUADS quads per round, all requested before the first is added: a lane's consecutive quads lie in one or two cache
            // lines, and requested together they are ONE trip to the L2 where one quad per round was a trip each -- the lines a wave
            // touches in a round, 64 lanes x 2 arrays, do not survive in the L1 until its next round)
            for (int x = x0; x <= x1; x += VU_STEP * VU_QUADS) {
                Quad<int> lab[VU_QUADS];
                Quad<float> val[VU_QUADS];
#pragma unroll
                for (int q = 0; q < VU_QUADS; ++q) {
                    lab[q] = { { -1, -1, -1, -1 } };
                    if (x + VU_STEP * q <= x1) lab[q] = *reinterpret_cast<const Quad<int> *>(labels + row + x + VU_STEP * q);
                }
#pragma unroll
                for (int q = 0; q < VU_QUADS; ++q) {
                    val[q] = { { 0.f, 0.f, 0.f, 0.f } };
#ifndef VOL_UPDATE_EAGER_VALUES
                    if (lab[q].v[0] == k || lab[q].v[1] == k || lab[q].v[2] == k || lab[q].v[3] == k)
#else
                    if (x + VU_STEP * q <= x1)
#endif
                        val[q] = *reinterpret_cast<const Quad<float> *>(vol + row + x + VU_STEP * q);
                }
#pragma unroll
                for (int q = 0; q < VU_QUADS; ++q)
#pragma unroll
                    for (int j = 0; j < VU_STEP; ++j) {
                        const int xx = x + VU_STEP * q + j;
                        const bool mine = lab[q].v[j] == k && xx <= x1;
                        sz = sz + (mine ? fz : 0.f);
                        sy = sy + (mine ? fy : 0.f);
                        sx = sx + (mine ? (float)xx : 0.f);
                        sv = sv + (mine ? val[q].v[j] : 0.f);
                        cnt += mine ? 1 : 0;
                    }
            }
        }
    }
    const float fc = (float)cnt;                              // seg[c] / (float)cnt   (cnt > 0: the box is not empty)
    const float cz = sz / fc, cy = sy / fc, cx = sx / fc;
    *reinterpret_cast<float4 *>(s.cen32 + (size_t)k * 4) = make_float4(cz, cy, cx, sv / fc);
    vol_window_f32(s, cz, cy, cx, w);
    vol_bbox_reset(bb);
}

int launch_vol_slic_f32(VolState s, const float *vol, int32_t *labels, int max_iter, hipStream_t st, const ProfHook *prof)
{
    size_t n = (size_t)s.D * s.H * s.W;
    HIP_TRY(hipMemsetAsync(labels, 0xff, n * sizeof(int32_t), st));
    hipLaunchKernelGGL(k_vol_centroid_init_f32, cdiv(s.K, 256), 256, 0, st, s);
    dim3 grid(cdiv(s.W, 64), cdiv(s.H, 4 * VROWS) * cdiv(s.D, VZ));      // a workgroup: 64 x 16 voxels of VZ slices
    const size_t n_bricks = (size_t)s.nbz * s.nby * s.nbx;
    for (int it = 0; it < max_iter; ++it) {
        HIP_TRY(hipMemsetAsync(s.brick_count, 0, n_bricks * sizeof(int), st));
        hipLaunchKernelGGL(k_vol_scatter_f32, cdiv((long)s.K * 64, 256), 256, 0, st, s);
        // (when profiling: the event pair rides on the dispatch of the assignment kernel, group 0 = "slic_assign")
        hipEvent_t ev_a = nullptr, ev_b = nullptr;
        if (prof && prof->pair) prof->pair(prof->user, 0, &ev_a, &ev_b);
        if (it + 1 < max_iter) {
            if (ev_a) hipExtLaunchKernelGGL(k_vol_assign_f32<true>, grid, dim3(256), 0, st, ev_a, ev_b, 0, s, vol, labels);
            else hipLaunchKernelGGL(k_vol_assign_f32<true>, grid, 256, 0, st, s, vol, labels);
            hipLaunchKernelGGL(k_vol_update_f32_lane, cdiv(s.K, VOL_UPDATE_BLOCK), VOL_UPDATE_BLOCK, 0, st, s, vol, labels);
        } else {
            if (ev_a) hipExtLaunchKernelGGL(k_vol_assign_f32<false>, grid, dim3(256), 0, st, ev_a, ev_b, 0, s, vol, labels);
            else hipLaunchKernelGGL(k_vol_assign_f32<false>, grid, 256, 0, st, s, vol, labels);
        }
    }
    HIP_TRY(hipGetLastError());
    return 0;
}

// ---- skimage.measure.label: full (26-/8-) connectivity, value 0 = background ------------------------------
__device__ __forceinline__ int cc_find(const int32_t *parent, int a)
{
    int p = parent[a];
    while (p != a) {
        a = p;
        p = parent[a];
    }
    return a;
}
__device__ __forceinline__ void cc_union(int32_t *parent, int a, int b)
{
    while (true) {
        a = cc_find(parent, a);
        b = cc_find(parent, b);
        if (a == b) return;
        if (a < b) {
            int t = a;
            a = b;
            b = t;
        }
        int old = atomicMin(&parent[a], b);
        if (old == a) return;
        a = old;
    }
}

// union with the two finds walked together (both loads of a step in flight at once: half the dependent trips of cc_union)
__device__ __forceinline__ void cc_union_pair(int32_t *parent, int a, int b)
{
    while (true) {
        while (true) {
            const int pa = parent[a], pb = parent[b];
            if (pa == a && pb == b) break;
            a = pa;
            b = pb;
        }
        if (a == b) return;
        if (a < b) {
            const int t = a;
            a = b;
            b = t;
        }
        const int old = atomicMin(&parent[a], b);
        if (old == a) return;
        a = old;
    }
}

__global__ void __launch_bounds__(256) k_cc_init(int32_t *parent, int n)
{
    int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p < n) parent[p] = p;
}

__global__ void __launch_bounds__(256)
k_cc_merge_full(const int32_t *__restrict__ labels, int32_t *parent, int D, int H, int W)
{
    int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= D * H * W) return;
    const int l = labels[p];
    if (l == 0) return;                              // background is never joined
    const int x = p % W, y = (p / W) % H, z = p / (W * H);
    // the 13 "earlier" neighbours of the full 3 x 3 x 3 neighbourhood
    for (int dz = -1; dz <= 0; ++dz)
        for (int dy = -1; dy <= 1; ++dy)
            for (int dx = -1; dx <= 1; ++dx) {
                if (dz == 0 && (dy > 0 || (dy == 0 && dx >= 0))) continue;
                int zz = z + dz, yy = y + dy, xx = x + dx;
                if (zz < 0 || yy < 0 || yy >= H || xx < 0 || xx >= W) continue;
                int q = (zz * H + yy) * W + xx;
                if (labels[q] == l) cc_union(parent, p, q);
            }
}

// Round 5 (k_cc_merge_runs, 28.9 ms at 2^30 voxels; replaced by k_cc_merge_rows below, which keeps its rule): the same components
// with a handful of unions per RUN instead of thirteen per voxel.  Every voxel ties itself to its left neighbour when the labels
// agree, so the voxels of a run (equal labels side by side in one row) are one set.  For each of the four earlier rows that touch
// p -- (z, y-1), (z-1, y-1), (z-1, y), (z-1, y+1) -- with a, b, c its voxels at x-1, x, x+1:
//   * p has no equal left neighbour (a run starts): b equal -> union with b (a and c, if equal, hang on b's run); else union with a
//     and with c, whichever is equal;
//   * p continues a run: its left neighbour is tied to its own equal neighbours of that row, which include a and b, so only c can
//     be news, and only when b is not equal (otherwise c hangs on b's run).
// By induction along the run every voxel ends up in one set with every equal voxel of its 26-neighbourhood, i.e. the components are
// those of k_cc_merge_full; the root of a set is its smallest index either way (cc_union), so numbering and result are identical.
// Unions happen where runs start or the row above changes -- on the surface of the segments, not in their volume.

// Round 6: the same merge rule with the five rows it looks at -- (z, y) and the four earlier rows -- loaded ONCE per wave and the
// x - 1 / x + 1 neighbours taken from the neighbouring lanes (one DPP move each) instead of up to thirteen loads per voxel, no
// division for the coordinates (grid = row segments, y, z), and the forest initialised by runs: a wave covers MR_SPAN = 62 voxels of
// a row with lane 0 and lane 63 carrying the voxels left and right of them; k_cc_init_rows points every voxel of a run at the
// run's first voxel INSIDE its segment, so the merge kernel ties a voxel to its left neighbour only where a run crosses into the
// segment.  Same sets, same roots (the smallest index of a set) as k_cc_merge_runs / k_cc_merge_full.
constexpr int MR_SPAN = 62;

__global__ void __launch_bounds__(256)
k_cc_init_rows(const int32_t *__restrict__ labels, int32_t *__restrict__ parent, int H, int W)
{
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int x = (blockIdx.x * 4 + wave) * MR_SPAN + lane - 1;
    const size_t row = ((size_t)blockIdx.z * H + blockIdx.y) * W;
    const bool inx = x >= 0 && x < W;
    const int l = inx ? labels[row + x] : -1;
    const bool mine = inx && lane >= 1 && lane <= MR_SPAN;
    const bool cont = lane_prev(l, -1) == l && lane > 1 && l != 0;          // continues a run that began inside this segment
    const unsigned long long starts = __ballot(mine && !cont);
    if (!mine) return;
    const unsigned long long below = starts & ((2ULL << lane) - 1ULL);     // (lane <= 62)
    const int start_lane = 63 - __clzll((long long)below);
    parent[row + x] = (int)(row + x) - (lane - start_lane);
}

// MR_ROWS rows of the slice per wave: the MR_ROWS + 1 rows of the slice and the MR_ROWS + 2 rows of the slice behind that they touch
// are loaded once, and the unions of a lane over its rows -- a bit each in `todo`: 13 r + 3 e + (dx + 1) for the voxel at x + dx of
// earlier row e, 13 r + 12 for the left neighbour -- are done two at a time (union2_min_root), every lane that still has some side by
// side.  (One row per wave, one union per lane and round: 16.4 ms at 2^30 voxels.)
constexpr int MR_ROWS = 4;

__global__ void __launch_bounds__(256)
k_cc_merge_rows(const int32_t *__restrict__ labels, int32_t *parent, int D, int H, int W)
{
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int x = (blockIdx.x * 4 + wave) * MR_SPAN + lane - 1;
    const int y0 = blockIdx.y * MR_ROWS, z = blockIdx.z;
    const bool inx = x >= 0 && x < W;
    const int plane = H * W;
    const size_t row0 = (size_t)z * plane + (size_t)y0 * W;
    // cz[i]: row y0 - 1 + i of this slice, pz[i]: row y0 - 1 + i of the slice behind; -1 where there is none (labels are >= 0)
    int cz[MR_ROWS + 1], pz[MR_ROWS + 2];
#pragma unroll
    for (int i = 0; i <= MR_ROWS; ++i) {
        const int y = y0 - 1 + i;
        cz[i] = (inx && y >= 0 && y < H) ? labels[row0 + (size_t)(i - 1) * W + x] : -1;
    }
#pragma unroll
    for (int i = 0; i <= MR_ROWS + 1; ++i) {
        const int y = y0 - 1 + i;
        pz[i] = (inx && z > 0 && y >= 0 && y < H) ? labels[row0 - plane + (size_t)(i - 1) * W + x] : -1;
    }
    int czp[MR_ROWS + 1], czn[MR_ROWS + 1], pzp[MR_ROWS + 2], pzn[MR_ROWS + 2];
#pragma unroll
    for (int i = 0; i <= MR_ROWS; ++i) {
        czp[i] = lane_prev(cz[i], -1);
        czn[i] = lane_next(cz[i], -1);
    }
#pragma unroll
    for (int i = 0; i <= MR_ROWS + 1; ++i) {
        pzp[i] = lane_prev(pz[i], -1);
        pzn[i] = lane_next(pz[i], -1);
    }
    const bool seg = inx && lane >= 1 && lane <= MR_SPAN;
    unsigned long long todo = 0;
#pragma unroll
    for (int r = 0; r < MR_ROWS; ++r) {
        const int l = cz[1 + r];
        const bool mine = seg && y0 + r < H && l != 0;                      // background is never joined
        const bool left = czp[1 + r] == l;
        if (mine && left && lane == 1) todo |= 1ULL << (13 * r + 12);        // a run that crosses into the segment
        // the four earlier rows that touch row y0 + r: (z, y-1), (z-1, y-1), (z-1, y), (z-1, y+1)
        const int el[4] = { cz[r], pz[r], pz[r + 1], pz[r + 2] };
        const int ep[4] = { czp[r], pzp[r], pzp[r + 1], pzp[r + 2] };
        const int en[4] = { czn[r], pzn[r], pzn[r + 1], pzn[r + 2] };
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const bool a = ep[e] == l, b = el[e] == l, c = en[e] == l;
            if (!mine) continue;
            if (left) {
                if (c && !b) todo |= 1ULL << (13 * r + 3 * e + 2);
            } else if (b) {
                todo |= 1ULL << (13 * r + 3 * e + 1);
            } else {
                if (a) todo |= 1ULL << (13 * r + 3 * e);
                if (c) todo |= 1ULL << (13 * r + 3 * e + 2);
            }
        }
    }
    const int p0 = (int)(row0 + x);
    while (__any(todo != 0)) {
        int ua[2] = { -1, -1 }, ub[2] = { -1, -1 };
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            if (!todo) continue;
            const int bit = __ffsll((long long)todo) - 1;
            todo &= todo - 1;
            const int r = bit / 13, k = bit - 13 * r;
            const int p = p0 + r * W;
            ua[j] = p;
            if (k == 12) {
                ub[j] = p - 1;
            } else {
                const int e = k / 3, dx = k - 3 * e - 1;
                ub[j] = p + dx + (e == 0 ? -W : e == 1 ? -plane - W : e == 2 ? -plane : -plane + W);
            }
        }
        union2_min_root(parent, ua[0], ub[0], ua[1], ub[1]);
    }
}

// four consecutive words of an int32 array (one 16-byte load where all four exist, `fill` behind the end)
__device__ __forceinline__ void load4_i32(const int32_t *a, int p, int n, int fill, int (&v)[4])
{
    if (p + 4 <= n) {
        const int4 q = *reinterpret_cast<const int4 *>(a + p);
        v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
    } else {
#pragma unroll
        for (int c = 0; c < 4; ++c) v[c] = p + c < n ? a[p + c] : fill;
    }
}

// roots of non-background components are numbered 1, 2, ... in raster order (block scan in three steps).  A workgroup takes
// CC_BLOCK voxels as CC_TILES tiles of 1 024 -- four consecutive voxels per lane, one 16-byte load, a wave reads 1 KB contiguous
// (16 consecutive voxels per lane, as before round 6, made every load instruction touch 64 cache lines: 2.3 ms a pass at 2^30
// voxels).  Only ROOTS are looked at -- parent[p] == p, which the merge pass leaves final -- so the forest is not flattened first:
// k_cc_write walks from every voxel to its root itself.
constexpr int CC_TILES = 4;
constexpr int CC_BLOCK = CC_TILES * 1024;

template <int NW> __device__ __forceinline__ int cc_block_scan(int v, int *total)
{
    __shared__ int wsum[NW];
    int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int incl = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        int t = __shfl_up(incl, off, 64);
        if (lane >= off) incl += t;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    int base = 0, all = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
        base += w < wave ? wsum[w] : 0;
        all += wsum[w];
    }
    *total = all;
    __syncthreads();
    return base + incl - v;
}

template <bool ASSIGN>
__global__ void __launch_bounds__(256)
k_cc_number(const int32_t *__restrict__ labels, const int32_t *__restrict__ parent, int n, int32_t *blocksum,
            int32_t *newlabel)
{
    const int base = blockIdx.x * CC_BLOCK + threadIdx.x * 4;
    unsigned fg = 0, roots = 0;                     // bit 4 * tile + c: voxel is the root of a foreground / of any component
    int cnt[CC_TILES];
#pragma unroll
    for (int i = 0; i < CC_TILES; ++i) {
        const int p = base + i * 1024;
        int v[4];
        load4_i32(parent, p, n, -1, v);
        cnt[i] = 0;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            if (v[c] != p + c) continue;
            roots |= 1u << (4 * i + c);
            if (labels[p + c] != 0) {
                fg |= 1u << (4 * i + c);
                cnt[i]++;
            }
        }
    }
    if (!ASSIGN) {
        int total;
        cc_block_scan<4>(cnt[0] + cnt[1] + cnt[2] + cnt[3], &total);
        if (threadIdx.x == 0) blocksum[blockIdx.x] = total;
    } else {
        int rank0 = blocksum[blockIdx.x];
#pragma unroll
        for (int i = 0; i < CC_TILES; ++i) {
            int total;
            int rank = rank0 + cc_block_scan<4>(cnt[i], &total);
            rank0 += total;
#pragma unroll
            for (int c = 0; c < 4; ++c)
                if (roots >> (4 * i + c) & 1u) newlabel[base + i * 1024 + c] = (fg >> (4 * i + c) & 1u) ? 1 + rank++ : 0;
        }
    }
}

__global__ void __launch_bounds__(1024) k_cc_scan_blocks(int32_t *blocksum, int nblocks, int32_t *total_out)
{
    __shared__ int carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int base = 0; base < nblocks; base += 1024) {
        int i = base + threadIdx.x;
        int v = i < nblocks ? blocksum[i] : 0;
        int total;
        int excl = cc_block_scan<16>(v, &total);
        if (i < nblocks) blocksum[i] = carry + excl;
        __syncthreads();
        if (threadIdx.x == 0) carry += total;
        __syncthreads();
    }
    if (threadIdx.x == 0) *total_out = carry;
}

// out[p] = number of p's root: four voxels per lane, their walks to the root side by side (four loads in flight per step; the
// forest is what the merge pass left -- a run's voxels point at its first voxel, that one at an earlier run -- two or three steps)
__global__ void __launch_bounds__(256)
k_cc_write(const int32_t *__restrict__ parent, const int32_t *__restrict__ newlabel, int n, int32_t *out)
{
    const int p = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (p >= n) return;
    int r[4];
    load4_i32(parent, p, n, 0, r);
    while (true) {
        int q[4];
        bool moved = false;
#pragma unroll
        for (int c = 0; c < 4; ++c) q[c] = parent[r[c]];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            moved |= q[c] != r[c];
            r[c] = q[c];
        }
        if (!moved) break;
    }
    int o[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) o[c] = newlabel[r[c]];
    if (p + 4 <= n) {
        *reinterpret_cast<int4 *>(out + p) = make_int4(o[0], o[1], o[2], o[3]);
    } else {
        for (int c = 0; p + c < n; ++c) out[p + c] = o[c];
    }
}

int launch_label_cc(int32_t *labels_inout, int D, int H, int W, int32_t *parent, int32_t *newlabel, int32_t *blocksum,
                    int32_t *total_dev, hipStream_t st)
{
    const int n = D * H * W, grid = cdiv(n, 256), nb = cdiv(n, CC_BLOCK);
    if (knobs().cc_merge_full || H > 65535 || D > 65535) {
        hipLaunchKernelGGL(k_cc_init, grid, 256, 0, st, parent, n);
        hipLaunchKernelGGL(k_cc_merge_full, grid, 256, 0, st, labels_inout, parent, D, H, W);
    } else {
        const dim3 rows(cdiv(W, 4 * MR_SPAN), H, D), row_groups(cdiv(W, 4 * MR_SPAN), cdiv(H, MR_ROWS), D);
        hipLaunchKernelGGL(k_cc_init_rows, rows, 256, 0, st, labels_inout, parent, H, W);
        hipLaunchKernelGGL(k_cc_merge_rows, row_groups, 256, 0, st, labels_inout, parent, D, H, W);
    }
    hipLaunchKernelGGL(k_cc_number<false>, nb, 256, 0, st, labels_inout, parent, n, blocksum, newlabel);
    hipLaunchKernelGGL(k_cc_scan_blocks, 1, 1024, 0, st, blocksum, nb, total_dev);
    hipLaunchKernelGGL(k_cc_number<true>, nb, 256, 0, st, labels_inout, parent, n, blocksum, newlabel);
    hipLaunchKernelGGL(k_cc_write, cdiv(cdiv(n, 4), 256), 256, 0, st, parent, newlabel, n, labels_inout);
    HIP_TRY(hipGetLastError());
    return 0;
}

// ---- 6-connected adjacency bitmap + centre sums of a label volume -----------------------------------------
// (table: instead of bit (row b, column a) of a K x K bitmap, every label goes into the row of each of its neighbours in a K x cap
// table of neighbour slots -- open addressing inside the row, -1 = free; a row that is full raises *overflow and the caller comes
// back with wider rows.  The bitmap is 11 GB for the 3 * 10^5 supervoxels of BASELINE configs[4] and caps K; the table is
// K * cap * 4 bytes.  Round 6: the table is SYMMETRIC (rounds 4 / 5 kept the smaller neighbours only), so that the fused call can
// build its arcs from it -- terms.hip k_tab_sort_rows / k_tab_emit -- as it does from the mirrored bitmap.)
__device__ __forceinline__ void neighbour_insert(int32_t *table, int cap, int b, int a, int *overflow)
{
    int32_t *row = table + (size_t)b * cap;
    unsigned slot = ((unsigned)a * 2654435761u) >> 7;
    for (int probe = 0; probe < cap; ++probe, ++slot) {
        int32_t *cell = row + (slot & (unsigned)(cap - 1));
        int seen = __hip_atomic_load(cell, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (seen == a) return;
        if (seen == -1) {
            seen = atomicCAS(cell, -1, a);
            if (seen == -1 || seen == a) return;
        }
    }
    *overflow = 1;
}

// Round 6: by rows and runs, like the 2-D kernel (graph.hip).  A wave owns 64 x VA_ROWS voxels of one slice -- VA_ROWS + 1 rows of it
// and VA_ROWS rows of the next slice in registers, the left / right neighbour through one DPP move each.
//   * neighbour pairs: across x a pair exists exactly where a run ends; across y and z the pair (label, label below / behind) of a
//     voxel is the pair of its left neighbour along the whole contact of two segments, so only the lane where EITHER label changes
//     reports it -- a handful of inserts per run instead of one per surface voxel;
//   * centre sums: the first lane of a run knows its length from the vote of the run starts, hence n, sum y, sum x of the run in
//     closed form (z is the workgroup's); they meet in an LDS hash table of the workgroup (a 64 x 16 cross-section sees a dozen
//     labels), flushed with one set of int64 global atomics per label and workgroup.
// Rounds 2 - 5 went voxel by voxel: two 32-bit divisions per voxel for its coordinates, a serial loop over the distinct labels of
// a wave with eight int64 wave reductions and four global atomics each (19.8 ms for the 2^30 voxels of BASELINE configs[4]).
// MODE 0: bit (row b, column a), a < b, of the K x K bitmap; MODE 1: a into row b AND b into row a of the neighbour table.
constexpr int VA_ROWS = 4;
constexpr int VA_SLOTS = 64;
constexpr int VA_DEPTH = 8;           // slices a workgroup walks (sums of a table slot stay far below 2^31: 8 192 voxels x 65 535)

// The (up to) three neighbour pairs a voxel reports, both directions each: the six table cells a pair's labels hash to are looked at
// TOGETHER -- one trip to memory -- and nearly always hold the label already (a pair of neighbouring supervoxels is reported by every
// voxel along their common face); what is not found there goes through neighbour_insert.  (One pair after the other, each with
// its own look: twenty-four dependent trips per wave of four rows, 4.6 ms for the 2^30 voxels of config 5.)
__device__ __forceinline__ void neighbour_insert3(int32_t *table, int cap, int l, const int (&nb)[3], int *overflow)
{
    int seen[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const int row = i < 3 ? l : nb[i - 3], val = i < 3 ? nb[i] : l;
        seen[i] = val;                                             // (no pair: nothing to do)
        if (nb[i % 3] >= 0)
            seen[i] = __hip_atomic_load(table + (size_t)row * cap + ((((unsigned)val * 2654435761u) >> 7) & (unsigned)(cap - 1)),
                                        __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const int row = i < 3 ? l : nb[i - 3], val = i < 3 ? nb[i] : l;
        if (nb[i % 3] >= 0 && seen[i] != val) neighbour_insert(table, cap, row, val, overflow);
    }
}

template <int MODE>
__device__ __forceinline__ void adjacency_report(int l, int nb, int words, uint32_t *bitmap, int32_t *table, int cap, int *overflow)
{
    if (MODE == 1) {
        neighbour_insert(table, cap, l, nb, overflow);
        neighbour_insert(table, cap, nb, l, overflow);
    } else {
        const int a = min(l, nb), b = max(l, nb);
        uint32_t *wp = bitmap + (size_t)b * words + (a >> 5);
        const uint32_t bit = 1u << (a & 31);
        if (!(*wp & bit)) atomicOr(wp, bit);
    }
}

template <int MODE>
__global__ void __launch_bounds__(256)
k_vol_adjacency_runs(const int32_t *__restrict__ labels, int D, int H, int W, int words, uint32_t *bitmap,
                     long long *__restrict__ cacc, int32_t *table, int cap, int *overflow)
{
    // (round 6, later: a workgroup walks VA_DEPTH slices -- the rows of the slice behind are the next turn's own rows, so a voxel is
    // loaded once instead of twice, and the centre sums of all the slices meet in one LDS table: 4.85 -> see profiles/README_r06.md)
    __shared__ int h_key[VA_SLOTS], h_n[VA_SLOTS], h_sz[VA_SLOTS], h_sy[VA_SLOTS], h_sx[VA_SLOTS];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (threadIdx.x < VA_SLOTS) {
        h_key[threadIdx.x] = -1;
        h_n[threadIdx.x] = 0;
        h_sz[threadIdx.x] = 0;
        h_sy[threadIdx.x] = 0;
        h_sx[threadIdx.x] = 0;
    }
    __syncthreads();
    const int z_first = blockIdx.z * VA_DEPTH, z_end = min(z_first + VA_DEPTH, D);
    const int y0 = (blockIdx.y * 4 + wave) * VA_ROWS;
    const int x = blockIdx.x * 64 + lane;
    const bool xin = x < W;
    const size_t plane = (size_t)H * W;
    const unsigned long long le = (lane == 63) ? ~0ULL : ((2ULL << lane) - 1ULL);
    // (edge: the voxel right of the wave's last lane, fetched by that lane WITH the rows -- asked for row by row where it is used, it
    // was a trip to memory per row and slice for the sake of one lane)
    int lab[VA_ROWS + 1], behind[VA_ROWS + 1], edge[VA_ROWS], edge_behind[VA_ROWS];
    const bool last = lane == 63 && x + 1 < W;
    {
        const int32_t *__restrict__ first = labels + ((size_t)z_first * H + y0) * W;
#pragma unroll
        for (int r = 0; r <= VA_ROWS; ++r) lab[r] = (xin && y0 + r < H) ? first[(size_t)r * W + x] : -1;
#pragma unroll
        for (int r = 0; r < VA_ROWS; ++r) edge[r] = (last && y0 + r < H) ? first[(size_t)r * W + x + 1] : -1;
#pragma unroll
        for (int r = 0; r <= VA_ROWS; ++r) behind[r] = (xin && y0 + r < H && z_first + 1 < D) ? first[plane + (size_t)r * W + x] : -1;
#pragma unroll
        for (int r = 0; r < VA_ROWS; ++r) edge_behind[r] = (last && y0 + r < H && z_first + 1 < D) ? first[plane + (size_t)r * W + x + 1] : -1;
    }
    for (int z = z_first; z < z_end; ++z) {
        // (the rows of slice z + 2 are requested HERE and used in the next turn: a turn does not wait for its own loads)
        const int32_t *__restrict__ base = labels + ((size_t)z * H + y0) * W;          // (wave uniform)
        const bool more = z + 1 < z_end && z + 2 < D;
        int ahead[VA_ROWS + 1], edge_ahead[VA_ROWS];
#pragma unroll
        for (int r = 0; r <= VA_ROWS; ++r) ahead[r] = (more && xin && y0 + r < H) ? base[2 * plane + (size_t)r * W + x] : -1;
#pragma unroll
        for (int r = 0; r < VA_ROWS; ++r) edge_ahead[r] = (more && last && y0 + r < H) ? base[2 * plane + (size_t)r * W + x + 1] : -1;
#pragma unroll
        for (int r = 0; r < VA_ROWS; ++r) {
            const int y = y0 + r;
            const int l = lab[r];
            const bool act = l >= 0;                                   // (the active lanes of a row are lanes 0 .. nact - 1)
            int right = lane_next(l, -1);
            if (lane == 63) right = edge[r];
            const int left = lane_prev(l, -2);
            const int below = lab[r + 1], back = behind[r];
            const int below_left = lane_prev(below, -2), back_left = lane_prev(back, -2);
            const bool rep_right = act && right >= 0 && right != l;
            const bool rep_below = act && below >= 0 && below != l && !(left == l && below_left == below);
            const bool rep_back = act && back >= 0 && back != l && !(left == l && back_left == back);
            if (MODE == 1) {
                if (rep_right || rep_below || rep_back) {
                    const int nb[3] = { rep_right ? right : -1, rep_below ? below : -1, rep_back ? back : -1 };
                    neighbour_insert3(table, cap, l, nb, overflow);
                }
            } else {
                if (rep_right) adjacency_report<MODE>(l, right, words, bitmap, table, cap, overflow);
                if (rep_below) adjacency_report<MODE>(l, below, words, bitmap, table, cap, overflow);
                if (rep_back) adjacency_report<MODE>(l, back, words, bitmap, table, cap, overflow);
            }
            const bool start = act && left != l;                       // (lane 0: left = -2)
            const unsigned long long starts = __ballot(start);
            const int nact = __popcll(__ballot(act));
            if (start) {
                const unsigned long long above = starts & ~le;
                const int len = (above ? __ffsll((long long)above) - 1 : nact) - lane;
                const int sy = len * y, sx = len * x + (len * (len - 1)) / 2;
                int slot = (int)(((unsigned int)l * 2654435761u) >> 26);          // 6 bits
                bool placed = false;
                for (int probe = 0; probe < VA_SLOTS; ++probe) {
                    const int old = atomicCAS(&h_key[slot], -1, l);
                    if (old == -1 || old == l) {
                        placed = true;
                        break;
                    }
                    slot = (slot + 1) & (VA_SLOTS - 1);
                }
                if (placed) {
                    atomicAdd(&h_n[slot], len);
                    atomicAdd(&h_sz[slot], len * z);
                    atomicAdd(&h_sy[slot], sy);
                    atomicAdd(&h_sx[slot], sx);
                } else {                                               // (more than VA_SLOTS labels in the slices of a 64 x 16 cross-section)
                    atomic_add_i64(cacc + (size_t)l * 4 + 0, len);
                    atomic_add_i64(cacc + (size_t)l * 4 + 1, (long long)len * z);
                    atomic_add_i64(cacc + (size_t)l * 4 + 2, (long long)sy);
                    atomic_add_i64(cacc + (size_t)l * 4 + 3, (long long)sx);
                }
            }
        }
#pragma unroll
        for (int r = 0; r <= VA_ROWS; ++r) {
            lab[r] = behind[r];
            behind[r] = ahead[r];
        }
#pragma unroll
        for (int r = 0; r < VA_ROWS; ++r) {
            edge[r] = edge_behind[r];
            edge_behind[r] = edge_ahead[r];
        }
    }
    __syncthreads();
    if (threadIdx.x < VA_SLOTS && h_key[threadIdx.x] >= 0) {
        const int k = h_key[threadIdx.x], n = h_n[threadIdx.x];
        atomic_add_i64(cacc + (size_t)k * 4 + 0, n);
        atomic_add_i64(cacc + (size_t)k * 4 + 1, h_sz[threadIdx.x]);
        atomic_add_i64(cacc + (size_t)k * 4 + 2, h_sy[threadIdx.x]);
        atomic_add_i64(cacc + (size_t)k * 4 + 3, h_sx[threadIdx.x]);
    }
}

__global__ void k_vol_centres_finalize(const long long *__restrict__ cacc, int K, double *centres, uint8_t *present)
{
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= K) return;
    long long n = cacc[(size_t)k * 4];
    present[k] = n > 0;
    for (int c = 0; c < 3; ++c)
        centres[3 * k + c] = n > 0 ? i64_to_double(cacc[(size_t)k * 4 + 1 + c]) / (double)n : -1.0;
}

static inline dim3 vol_adjacency_grid(int D, int H, int W) { return dim3(cdiv(W, 64), cdiv(H, 4 * VA_ROWS), cdiv(D, VA_DEPTH)); }

int launch_vol_adjacency(const int32_t *labels, int D, int H, int W, int K, int words, uint32_t *bitmap, long long *cacc,
                         double *centres, uint8_t *present, hipStream_t st)
{
    if (D > 65535 || cdiv(H, 4 * VA_ROWS) > 65535) {
        set_error("adjacency: more than 65 535 slices or 1 048 560 rows");
        return -1;
    }
    HIP_TRY(hipMemsetAsync(bitmap, 0, (size_t)K * words * sizeof(uint32_t), st));
    HIP_TRY(hipMemsetAsync(cacc, 0, (size_t)K * 4 * sizeof(long long), st));
    hipLaunchKernelGGL(k_vol_adjacency_runs<0>, vol_adjacency_grid(D, H, W), 256, 0, st, labels, D, H, W, words, bitmap, cacc, nullptr, 0, nullptr);
    hipLaunchKernelGGL(k_vol_centres_finalize, cdiv(K, 256), 256, 0, st, cacc, K, centres, present);
    HIP_TRY(hipGetLastError());
    return 0;
}

// the same with the neighbour table (K x cap slots, cap a power of two) instead of the bitmap; *overflow (device) is raised when a
// row was too narrow
int launch_vol_adjacency_table(const int32_t *labels, int D, int H, int W, int K, int32_t *table, int cap, int *overflow, long long *cacc,
                               double *centres, uint8_t *present, hipStream_t st)
{
    if (D > 65535 || cdiv(H, 4 * VA_ROWS) > 65535) {
        set_error("adjacency: more than 65 535 slices or 1 048 560 rows");
        return -1;
    }
    HIP_TRY(hipMemsetAsync(table, 0xff, (size_t)K * cap * sizeof(int32_t), st));
    HIP_TRY(hipMemsetAsync(overflow, 0, sizeof(int), st));
    HIP_TRY(hipMemsetAsync(cacc, 0, (size_t)K * 4 * sizeof(long long), st));
    hipLaunchKernelGGL(k_vol_adjacency_runs<1>, vol_adjacency_grid(D, H, W), 256, 0, st, labels, D, H, W, 0, nullptr, cacc, table, cap, overflow);
    hipLaunchKernelGGL(k_vol_centres_finalize, cdiv(K, 256), 256, 0, st, cacc, K, centres, present);
    HIP_TRY(hipGetLastError());
    return 0;
}

#ifdef IMSEGM_VOL_PHASE_PROF
int vol_phase_read(unsigned long long *out16, int reset)
{
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_vol_phase), 16 * sizeof(unsigned long long)));
    if (reset) {
        unsigned long long zero[16] = { 0 };
        HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_vol_phase), zero, sizeof(zero)));
    }
    return 0;
}
#endif

}  // namespace imsegm

#ifdef IMSEGM_VOL_PHASE_PROF
extern "C" __attribute__((visibility("default"))) int imsegm_debug_vol_phases(unsigned long long *out16, int reset)
{
    return imsegm::vol_phase_read(out16, reset);
}
#endif
