// slic_pre.hip -- what runs in front of the SLIC sweeps: min / max of the image, and the pre-processing of
// skimage.segmentation.slic (img_as_float, rgb2lab, the sigma = 1 Gaussian, image / compactness) as ONE kernel through an LDS tile
// (k_pre_fused) or as three axis passes (larger radii).  Replaces the front half of the native boundary
// `skimage.segmentation.slic` called at /root/reference/imsegm/superpixels.py:61-63.  Arithmetic contract: bit-identical to
// oracle/imsegm_oracle.c (orc_slic_preprocess_color2d).  (Split off slic.hip in round 6.)
#include "slic.h"
#include <hip/hip_ext.h>
#include <atomic>
#include <cstdio>


namespace imsegm {

// ---------------------------------------------------------------------------------------------
// min / max of the input (superpixels.py:53-54), order-preserving uint64 keys + atomics
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned long long f64_key(double x)
{
    long long b = __double_as_longlong(x);
    return b < 0 ? ~(unsigned long long)b : ((unsigned long long)b | 0x8000000000000000ULL);
}
__host__ __device__ __forceinline__ double key_f64(unsigned long long k)
{
    unsigned long long b = (k & 0x8000000000000000ULL) ? (k & 0x7fffffffffffffffULL) : ~k;
    union { unsigned long long u; double d; } cv;
    cv.u = b;
    return cv.d;
}

// wave reduce, block reduce through LDS, then ONE atomic pair per block (a per-wave atomic on the
// same two words serialises at ~12 ns each).  The words rest at zero between calls: keys[0] holds the complement of the
// minimum's key (so both are running maxima and zero is the neutral element), keys[5] is an arrival counter (the call's results sit in between), and the
// workgroup that arrives last decodes the pair into out2, clears `also_zero` (the accumulator of the NEXT kernel of the
// stream) and puts the words back to zero -- no launch before or after the reduction.
__device__ __forceinline__ void block_minmax_commit(double mn, double mx, unsigned long long *keys, double *out2, double *also_zero)
{
    __shared__ double smn[4], smx[4];
    mn = wave_min_f64(mn);
    mx = wave_max_f64(mx);
    if ((threadIdx.x & 63) == 0) {
        smn[threadIdx.x >> 6] = mn;
        smx[threadIdx.x >> 6] = mx;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        mn = fmin(fmin(smn[0], smn[1]), fmin(smn[2], smn[3]));
        mx = fmax(fmax(smx[0], smx[1]), fmax(smx[2], smx[3]));
        __hip_atomic_fetch_max(&keys[0], ~f64_key(mn), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_max(&keys[1], f64_key(mx), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned *ticket = reinterpret_cast<unsigned *>(keys + 5);
        // (release / acquire at agent scope: the atomics above are performed before the ticket is taken, and the last
        // arriver's loads below see every workgroup's)
        const unsigned t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (t == gridDim.x - 1) {
            const unsigned long long k0 = __hip_atomic_exchange(&keys[0], 0ULL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned long long k1 = __hip_atomic_exchange(&keys[1], 0ULL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            out2[0] = key_f64(~k0);
            out2[1] = key_f64(k1);
            if (also_zero) *also_zero = 0.0;
        }
    }
}

template <typename T>
__global__ void __launch_bounds__(256) k_minmax(const T *__restrict__ src, size_t n, unsigned long long *keys, double *out2, double *also_zero,
                                                size_t zs)
{
    ZSHIFT(src, zs); ZSHIFT(keys, zs); ZSHIFT(out2, zs); ZSHIFT(also_zero, zs);
    double mn = INFINITY, mx = -INFINITY;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    // four independent loads in flight per lane (a 4.3 GB volume read one 4-byte load at a time ran at 0.33 TB/s)
    for (; i + 3 * stride < n; i += 4 * stride) {
        const double v0 = (double)src[i], v1 = (double)src[i + stride], v2 = (double)src[i + 2 * stride], v3 = (double)src[i + 3 * stride];
        mn = fmin(fmin(mn, v0), fmin(fmin(v1, v2), v3));
        mx = fmax(fmax(mx, v0), fmax(fmax(v1, v2), v3));
    }
    for (; i < n; i += stride) {
        const double v = (double)src[i];
        mn = fmin(mn, v);
        mx = fmax(mx, v);
    }
    block_minmax_commit(mn, mx, keys, out2, also_zero);
}

// uint8 variant: 16 bytes per lane
__global__ void __launch_bounds__(256) k_minmax_u8(const uint8_t *__restrict__ src, size_t n, unsigned long long *keys, double *out2,
                                                   double *also_zero, size_t zs)
{
    ZSHIFT(src, zs); ZSHIFT(keys, zs); ZSHIFT(out2, zs); ZSHIFT(also_zero, zs);
    unsigned mn = 255, mx = 0;
    size_t nvec = n / 16;
    const uint4 *v4 = reinterpret_cast<const uint4 *>(src);
    size_t stride = (size_t)gridDim.x * blockDim.x;
    // bytes 0 / 2 and 1 / 3 of a word as two 16-bit lanes each: packed 16-bit min / max, four bytes per instruction pair
    typedef unsigned short us2 __attribute__((ext_vector_type(2)));
    us2 pmn = { 255, 255 }, pmx = { 0, 0 };
    // four independent 16-byte loads per thread and round (few workgroups: the loads of a thread must overlap)
    for (size_t i0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i0 < nvec; i0 += 4 * stride) {
        uint4 q[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const size_t i = i0 + u * stride;
            q[u] = i < nvec ? v4[i] : v4[i0];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const unsigned w[4] = { q[u].x, q[u].y, q[u].z, q[u].w };
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const unsigned e = w[j] & 0x00ff00ffu, o = (w[j] >> 8) & 0x00ff00ffu;
                const us2 ev = __builtin_bit_cast(us2, e), ov = __builtin_bit_cast(us2, o);
                pmn = __builtin_elementwise_min(pmn, __builtin_elementwise_min(ev, ov));
                pmx = __builtin_elementwise_max(pmx, __builtin_elementwise_max(ev, ov));
            }
        }
    }
    mn = min((unsigned)pmn.x, (unsigned)pmn.y);
    mx = max((unsigned)pmx.x, (unsigned)pmx.y);
    if (blockIdx.x == 0)
        for (size_t i = nvec * 16 + threadIdx.x; i < n; i += blockDim.x) {
            unsigned v = src[i];
            mn = min(mn, v);
            mx = max(mx, v);
        }
    block_minmax_commit((double)mn, (double)mx, keys, out2, also_zero);
}

// `keys`: words 0, 1 and 5 are ZERO when the call is enqueued and zero again when it has run (the session zeroes them once, where it
// allocates them); `also_zero`: one more double the last workgroup clears (nullptr: none)
int launch_minmax(const void *src, int dtype, size_t n, unsigned long long *keys, double *out2, hipStream_t st, double *also_zero, ZBatch zb)
{
    // few workgroups: each ends with two atomics on the same two words, and single-lane atomics on one address serialise at
    // ~12 ns (512 workgroups: 16 us, 2048: 50 us for a 12.6 MB image that streams in 3 us)
    int gx = (int)std::min<size_t>(128, (n + 256 * 16 - 1) / (256 * 16));
    if (gx < 1) gx = 1;
    // (a volume: the commit of 2 048 workgroups is 50 us beside milliseconds of streaming -- eight workgroups per CU)
    if (n >= ((size_t)1 << 26)) gx = 2048;
    const dim3 grid(gx, 1, zb.nz);
    if (dtype == DT_U8)
        hipLaunchKernelGGL(k_minmax_u8, grid, 256, 0, st, (const uint8_t *)src, n, keys, out2, also_zero, zb.zs);
    else if (dtype == DT_F32)
        hipLaunchKernelGGL(k_minmax<float>, grid, 256, 0, st, (const float *)src, n, keys, out2, also_zero, zb.zs);
    else
        hipLaunchKernelGGL(k_minmax<double>, grid, 256, 0, st, (const double *)src, n, keys, out2, also_zero, zb.zs);
    HIP_TRY(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------------------------
// pre-processing pass 1: normalise -> rgb2lab -> z-axis blur tap (depth 1) -> planar fp64
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ double zblur_point(double v, const Taps &tz)
{
    // scipy correlate1d on a length-1 axis with 'reflect': every neighbour equals v
    if (tz.r < 0) return v;
    double tmp = v * tz.w[0];
    for (int j = tz.r; j >= 1; --j) tmp += (v + v) * tz.w[j];
    return tmp;
}

// the same sum with the tap loop unrolled over the largest radius the fused kernel takes: the taps (kernel arguments) become scalar
// registers loaded once, instead of one scalar load -- and one wait for it -- per tap, channel and pixel
template <int MAXR>
__device__ __forceinline__ double zblur_point_unrolled(double v, const Taps &tz)
{
    if (tz.r < 0) return v;
    double tmp = v * tz.w[0];
    const double vv = v + v;
#pragma unroll
    for (int j = MAXR; j >= 1; --j)
        if (j <= tz.r) tmp += vv * tz.w[j];
    return tmp;
}

// uint8 input: the sRGB linearisation collapses to a 256-entry table (built per block in LDS with
// the same det_pow24 as the per-pixel path).  minmax = {vmin, vmax} on device.
__global__ void __launch_bounds__(256)
k_pre_lab_u8(const uint8_t *__restrict__ img, int n, int normalize, const double *__restrict__ minmax,
             Taps tz, double *__restrict__ out)
{
    __shared__ double lut[256];
    {
        int v = threadIdx.x;
        double x;
        const bool norm = normalize == 1 || (normalize == 2 && (minmax[0] != 0.0 || minmax[1] != 1.0));
        if (norm) {
            double vmin = minmax[0], range = minmax[1] - minmax[0];
            x = (double)(uint8_t)(v - (int)vmin) / range;
        } else {
            x = (double)v * (1.0 / 255);
        }
        lut[v] = (x > 0.04045) ? det_pow24((x + 0.055) / 1.055) : x / 12.92;
    }
    __syncthreads();
    int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    double lin0 = lut[img[3 * (size_t)p + 0]], lin1 = lut[img[3 * (size_t)p + 1]], lin2 = lut[img[3 * (size_t)p + 2]];
    double X = lin0 * 0.412453 + lin1 * 0.357580 + lin2 * 0.180423;
    double Y = lin0 * 0.212671 + lin1 * 0.715160 + lin2 * 0.072169;
    double Z = lin0 * 0.019334 + lin1 * 0.119193 + lin2 * 0.950227;
    double f[3] = { X / 0.95047, Y / 1.0, Z / 1.08883 };
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        double t = f[c];
        f[c] = (t > 0.008856) ? det_cbrt(t) : 7.787 * t + 16.0 / 116.0;
    }
    double L = (116.0 * f[1]) - 16.0;
    double A = 500.0 * (f[0] - f[1]);
    double B = 200.0 * (f[1] - f[2]);
    out[p] = zblur_point(L, tz);
    out[(size_t)n + p] = zblur_point(A, tz);
    out[2 * (size_t)n + p] = zblur_point(B, tz);
}

template <typename T>
__global__ void __launch_bounds__(256)
k_pre_lab_f(const T *__restrict__ img, int n, int normalize, const double *__restrict__ minmax, Taps tz,
            double *__restrict__ out)
{
    int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    double rgb[3];
    const bool norm = normalize == 1 || (normalize == 2 && (minmax[0] != 0.0 || minmax[1] != 1.0));
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        double v = (double)img[3 * (size_t)p + c];
        if (norm) v = (v - minmax[0]) / (minmax[1] - minmax[0]);
        rgb[c] = v;
    }
    double L, A, B;
    rgb2lab_px(rgb[0], rgb[1], rgb[2], L, A, B);
    out[p] = zblur_point(L, tz);
    out[(size_t)n + p] = zblur_point(A, tz);
    out[2 * (size_t)n + p] = zblur_point(B, tz);
}

// block maximum of a non-negative double -> one atomicMax on its bit pattern (non-negative doubles order like
// unsigned integers); `out` must have been zeroed
__device__ __forceinline__ void block_absmax_to(double v, double *out)
{
    __shared__ unsigned long long wave_max[16];
    unsigned long long b = (unsigned long long)__double_as_longlong(v);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        unsigned long long o = __shfl_xor(b, off, 64);
        b = o > b ? o : b;
    }
    const int wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    if ((threadIdx.x & 63) == 0) wave_max[wave] = b;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int i = 1; i < nw; ++i) b = wave_max[i] > b ? wave_max[i] : b;
        if (b) atomicMax(reinterpret_cast<unsigned long long *>(out), b);
    }
}

__global__ void __launch_bounds__(256) k_absmax_f64(const double *__restrict__ src, size_t n, double *out)
{
    double m = 0.0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) m = fmax(m, fabs(src[i]));
    block_absmax_to(m, out);
}

int launch_absmax_f64(const double *src, size_t n, double *out_dev, hipStream_t st)
{
    HIP_TRY(hipMemsetAsync(out_dev, 0, sizeof(double), st));
    int grid = (int)std::min<size_t>(2048, (n + 256 * 8 - 1) / (256 * 8));
    hipLaunchKernelGGL(k_absmax_f64, grid < 1 ? 1 : grid, 256, 0, st, src, n, out_dev);
    HIP_TRY(hipGetLastError());
    return 0;
}

__device__ __forceinline__ int reflect_idx(int i, int n)
{
    if ((unsigned)i < (unsigned)n) return i;       // (inside: all but the pixels of the border tiles; the modulo below is ~30 instructions)
    if (n == 1) return 0;
    int p = 2 * n;
    i %= p;
    if (i < 0) i += p;
    if (i >= n) i = p - 1 - i;
    return i;
}

// pass 2 / 3: one scipy correlate1d pass along y (axis = 0) or x (axis = 1) of each [H][W] plane;
// the x pass also applies the final `image * (1 / compactness)`.
template <int AXIS>
__global__ void __launch_bounds__(256)
k_blur_axis(const double *__restrict__ src, double *__restrict__ dst, int H, int W, Taps t, double ratio, int scale)
{
    int x = blockIdx.x * 64 + (threadIdx.x & 63);
    int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    int plane = blockIdx.z;
    if (x >= W || y >= H) return;
    const double *s = src + (size_t)plane * H * W;
    double v;
    if (t.r < 0) {
        v = s[(size_t)y * W + x];
    } else {
        v = s[(size_t)y * W + x] * t.w[0];
        for (int j = t.r; j >= 1; --j) {
            double a, b;
            if (AXIS == 0) {
                a = s[(size_t)reflect_idx(y - j, H) * W + x];
                b = s[(size_t)reflect_idx(y + j, H) * W + x];
            } else {
                a = s[(size_t)y * W + reflect_idx(x - j, W)];
                b = s[(size_t)y * W + reflect_idx(x + j, W)];
            }
            v += (a + b) * t.w[j];
        }
    }
    if (scale) v = v * ratio;
    dst[(size_t)plane * H * W + (size_t)y * W + x] = v;
}

// ---------------------------------------------------------------------------------------------
// fused pre-processing: the three passes above in one kernel (same operations in the same order, so the
// planes are bit-identical), 27 B/px of HBM traffic instead of 96.  A workgroup produces a 64 x 16 tile:
// Lab (+ z tap) of the tile and its blur halo goes to LDS (the halo is converted redundantly, x1.7), the y
// pass runs LDS -> LDS one channel at a time, the x pass LDS -> HBM with the final 1/compactness.
// ---------------------------------------------------------------------------------------------
// (tile geometry overridable at compile time for A/B builds -- tools/variants_k.sh; e.g. -DSLIC_PF_TX=32 -DSLIC_PF_TY=32 converts
// 40 x 40 pixels per 32 x 32 outputs, x1.56 instead of x1.69, in 48 KB of LDS: 125.8 against 129.0 us)
#ifndef SLIC_PF_TX
#define SLIC_PF_TX 64
#endif
#ifndef SLIC_PF_TY
#define SLIC_PF_TY 16
#endif
constexpr int PF_TX = SLIC_PF_TX, PF_TY = SLIC_PF_TY, PF_MAXR = 8, PF_THREADS = 512;
static_assert(PF_THREADS % PF_TX == 0 && PF_TY % (PF_THREADS / PF_TX) == 0, "x pass: whole rows per pass");

template <typename T>
__global__ void __launch_bounds__(PF_THREADS)
k_pre_fused(const T *__restrict__ img, int H, int W, int normalize, const double *__restrict__ minmax, Taps tz, Taps ty,
            Taps tx, double ratio, double *__restrict__ out, double *premax, size_t zs)
{
    ZSHIFT(img, zs); ZSHIFT(minmax, zs); ZSHIFT(out, zs); ZSHIFT(premax, zs);
    double vmax = 0.0;                                    // max |value written| by this thread
    extern __shared__ double pf_sm[];
    __shared__ double lut[256];
    const int tid = threadIdx.x;
    const int ry = ty.r < 0 ? 0 : ty.r, rx = tx.r < 0 ? 0 : tx.r;
    const int tw = PF_TX + 2 * rx, th = PF_TY + 2 * ry;
    const unsigned int tw_magic = (unsigned int)((0x100000000ull + tw - 1) / tw);     // exact i / tw for i < 2^16
    double *L1 = pf_sm;                                   // [3][th][tw]  Lab + z tap
    double *L2 = pf_sm + (size_t)3 * th * tw;             // [PF_TY][tw]  y-blurred, one channel at a time
    const bool norm = normalize == 1 || (normalize == 2 && (minmax[0] != 0.0 || minmax[1] != 1.0));
    if (sizeof(T) == 1) {
        // uint8 input: the sRGB linearisation collapses to a 256-entry table
        if (tid < 256) {
            const int v = tid;
            double x;
            if (norm) {
                double vmin = minmax[0], range = minmax[1] - minmax[0];
                x = (double)(uint8_t)(v - (int)vmin) / range;
            } else {
                x = (double)v * (1.0 / 255);
            }
            lut[v] = (x > 0.04045) ? det_pow24((x + 0.055) / 1.055) : x / 12.92;
        }
        __syncthreads();
    }
    const int x0 = blockIdx.x * PF_TX, y0 = blockIdx.y * PF_TY;
    const size_t plane = (size_t)H * W;
    for (int i = tid; i < th * tw; i += PF_THREADS) {
        const int ly = (int)__umulhi((unsigned int)i, tw_magic), lx = i - ly * tw;
        const int gy = reflect_idx(y0 + ly - ry, H), gx = reflect_idx(x0 + lx - rx, W);
        const size_t p = (size_t)gy * W + gx;
        double L, A, B;
        if (sizeof(T) == 1) {
            double lin0 = lut[(int)img[3 * p + 0]], lin1 = lut[(int)img[3 * p + 1]], lin2 = lut[(int)img[3 * p + 2]];
            double X = lin0 * 0.412453 + lin1 * 0.357580 + lin2 * 0.180423;
            double Y = lin0 * 0.212671 + lin1 * 0.715160 + lin2 * 0.072169;
            double Z = lin0 * 0.019334 + lin1 * 0.119193 + lin2 * 0.950227;
            // (uint8 pixels: X, Z are 0 or >= 5e-5 -- the exact three-instruction quotient; Y / 1.0 = Y)
            double f[3] = { DIV_CONST_IN_RANGE(X, 0.95047), Y, DIV_CONST_IN_RANGE(Z, 1.08883) };
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                double t = f[c];
                f[c] = (t > 0.008856) ? det_cbrt(t) : 7.787 * t + 16.0 / 116.0;
            }
            L = (116.0 * f[1]) - 16.0;
            A = 500.0 * (f[0] - f[1]);
            B = 200.0 * (f[1] - f[2]);
        } else {
            double rgb[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                double v = (double)img[3 * p + c];
                if (norm) v = (v - minmax[0]) / (minmax[1] - minmax[0]);
                rgb[c] = v;
            }
            rgb2lab_px(rgb[0], rgb[1], rgb[2], L, A, B);
        }
        L1[i] = zblur_point_unrolled<PF_MAXR>(L, tz);
        L1[th * tw + i] = zblur_point_unrolled<PF_MAXR>(A, tz);
        L1[2 * th * tw + i] = zblur_point_unrolled<PF_MAXR>(B, tz);
    }
    __syncthreads();
    const int ox = tid % PF_TX, oy0 = tid / PF_TX;
    for (int c = 0; c < 3; ++c) {
        const double *s1 = L1 + (size_t)c * th * tw;
        for (int i = tid; i < PF_TY * tw; i += PF_THREADS) {
            const int oy = (int)__umulhi((unsigned int)i, tw_magic), lx = i - oy * tw;
            const double *col = s1 + (oy + ry) * tw + lx;
            double v;
            if (ty.r < 0) {
                v = col[0];
            } else {
                v = col[0] * ty.w[0];
                // (unrolled over the largest radius: the taps become scalar registers loaded once)
#pragma unroll
                for (int j = PF_MAXR; j >= 1; --j)
                    if (j <= ty.r) v += (col[-j * tw] + col[j * tw]) * ty.w[j];
            }
            L2[i] = v;
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < PF_TY / (PF_THREADS / PF_TX); ++q) {
            const int oy = oy0 + (PF_THREADS / PF_TX) * q;
            const double *row = L2 + oy * tw + ox + rx;
            double v;
            if (tx.r < 0) {
                v = row[0];
            } else {
                v = row[0] * tx.w[0];
#pragma unroll
                for (int j = PF_MAXR; j >= 1; --j)
                    if (j <= tx.r) v += (row[-j] + row[j]) * tx.w[j];
            }
            v = v * ratio;
            const int gy = y0 + oy, gx = x0 + ox;
            if (gy < H && gx < W) {
                out[(size_t)c * plane + (size_t)gy * W + gx] = v;
                vmax = fmax(vmax, fabs(v));
            }
        }
        __syncthreads();
    }
    block_absmax_to(vmax, premax);
}

int launch_preprocess_color2d(const void *img, int dtype, int H, int W, int normalize, const double *minmax_dev,
                              const Taps &tz, const Taps &ty, const Taps &tx, double ratio, double *bufA,
                              double *bufB, double *premax_dev, hipStream_t st, bool premax_zeroed, ZBatch zb)
{
    int n = H * W;
    int grid = cdiv(n, 256);
    if (tz.r <= PF_MAXR && ty.r <= PF_MAXR && tx.r <= PF_MAXR && !knobs().pre_3pass) {
        const int ry = ty.r < 0 ? 0 : ty.r, rx = tx.r < 0 ? 0 : tx.r;
        const size_t lds = ((size_t)3 * (PF_TY + 2 * ry) + PF_TY) * (PF_TX + 2 * rx) * sizeof(double);
        const void *fn = dtype == DT_U8 ? (const void *)k_pre_fused<uint8_t>
                       : dtype == DT_F32 ? (const void *)k_pre_fused<float> : (const void *)k_pre_fused<double>;
        // the opt-in above 48 KB is per device: remembered per (device, dtype); (benign race: the attribute only ever grows)
        static size_t lds_set[IMSEGM_MAX_DEVICES][3];
        int dev = 0;
        HIP_TRY(hipGetDevice(&dev));
        if (dev < 0 || dev >= IMSEGM_MAX_DEVICES || lds_set[dev][dtype] < lds) {
            HIP_TRY(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            if (dev >= 0 && dev < IMSEGM_MAX_DEVICES) lds_set[dev][dtype] = lds;
        }
        dim3 gf(cdiv(W, PF_TX), cdiv(H, PF_TY), zb.nz);
        if (!premax_zeroed) {
            if (zb.nz > 1) {
                set_error("preprocess: a batch needs its premax words zeroed by the caller");
                return -1;
            }
            HIP_TRY(hipMemsetAsync(premax_dev, 0, sizeof(double), st));
        }
        if (dtype == DT_U8)
            hipLaunchKernelGGL(k_pre_fused<uint8_t>, gf, PF_THREADS, lds, st, (const uint8_t *)img, H, W, normalize, minmax_dev, tz, ty, tx,
                               ratio, bufA, premax_dev, zb.zs);
        else if (dtype == DT_F32)
            hipLaunchKernelGGL(k_pre_fused<float>, gf, PF_THREADS, lds, st, (const float *)img, H, W, normalize, minmax_dev, tz, ty, tx,
                               ratio, bufA, premax_dev, zb.zs);
        else
            hipLaunchKernelGGL(k_pre_fused<double>, gf, PF_THREADS, lds, st, (const double *)img, H, W, normalize, minmax_dev, tz, ty, tx,
                               ratio, bufA, premax_dev, zb.zs);
        HIP_TRY(hipGetLastError());
        return 0;   // result in bufA
    }
    if (zb.nz > 1) {
        set_error("preprocess: the three-pass path (blur radius > 8) does not take a batch");
        return -1;
    }
    if (dtype == DT_U8)
        hipLaunchKernelGGL(k_pre_lab_u8, grid, 256, 0, st, (const uint8_t *)img, n, normalize, minmax_dev, tz, bufA);
    else if (dtype == DT_F32)
        hipLaunchKernelGGL(k_pre_lab_f<float>, grid, 256, 0, st, (const float *)img, n, normalize, minmax_dev, tz, bufA);
    else
        hipLaunchKernelGGL(k_pre_lab_f<double>, grid, 256, 0, st, (const double *)img, n, normalize, minmax_dev, tz, bufA);
    dim3 g(cdiv(W, 64), cdiv(H, 4), 3);
    hipLaunchKernelGGL(k_blur_axis<0>, g, 256, 0, st, bufA, bufB, H, W, ty, ratio, 0);
    hipLaunchKernelGGL(k_blur_axis<1>, g, 256, 0, st, bufB, bufA, H, W, tx, ratio, 1);
    HIP_TRY(hipGetLastError());
    return launch_absmax_f64(bufA, (size_t)3 * n, premax_dev, st);   // result in bufA
}

}  // namespace imsegm
