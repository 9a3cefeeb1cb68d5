// stats.hip -- per-superpixel colour statistics as segmented reductions (wave64 shuffles + LDS).
//
// Replaces the native boundary /root/reference/imsegm/features_cython.pyx:59-141
// (normColorFeatures, computeColorImage2dMean / Energy / Variance) and :144-219 (gray 3D).
// Semantics mirrored from the Cython code: the image is staged as float32
// (descriptors.py:233,261,293), `val * val` and `(img - mean)^2` are evaluated in float32 and
// accumulated in 64-bit; labels without pixels keep 0.
//
// The reference adds in raster order into an fp64 scalar.  Here every float32 term is converted to
// an exact two-limb fixed-point integer and summed with integer adds (associative), so the result
// does not depend on the reduction tree or on atomic ordering; it equals the exactly rounded sum,
// which differs from the reference's running fp64 sum by at most a few ulp (bit-identical whenever
// the terms are integers, e.g. uint8 images: mean and energy).
#include "slic.h"

namespace imsegm {

constexpr int ST_ROWS = 16;       // rows per lane: a wave covers 64 x 16 pixels, a workgroup 64 x 64 (a 16-lane row sums 256 pixels:
                                  // the exactness bound pow2_scale is built for)
constexpr int ST_SLOTS = 64;      // LDS hash slots (distinct labels per workgroup tile)

// fixed point with a caller-chosen scale: v * 2^sh = hi + lo * 2^-32
__device__ __forceinline__ void fix_split_sh(double v, double scale, long long &hi, long long &lo)
{
    double t = v * scale;
    long long h = (long long)t;
    double r = t - (double)h;
    hi = h;
    lo = (long long)(r * 4294967296.0);
}

template <typename T> __device__ __forceinline__ float load_f32(const T *p, size_t i) { return (float)p[i]; }

struct StatParams {
    int H, W, K;
    double scale_v, scale_e;      // 2^shift for the value / squared-value sums
    int planar;                   // 0: H x W x 3 interleaved, 1: three planes `plane_stride` elements apart
    size_t plane_stride;          //    (0: one gray plane -- the single-channel kernel, NC = 1)
    size_t n_pixels;              // H * W (guards the 4-byte pixel load of interleaved uint8 images)
    int u8_int;                   // uint8 image, power-of-two scales >= 1: integer block sums in the first pass
    int prescale;                 // 1: value = (raw * mul) / div before the float32 staging
    double mul, div;              //    (descriptors.py:1094 `(response * (log(1 + norm) / 0.03)) / norm`)
    const double *ssq_dev;        // prescale == 2: norm = sqrt(*ssq_dev) on the device, mul and div derived from it (0 / inf norm: all values 0)
    size_t zs;                    // several images per launch (ZBatch): image blockIdx.z, every buffer zs bytes further on per image
};

// PASS 1 -> n + 3 x (v, v*v); PASS 2 -> 3 x (v - m)^2.
// One pixel column per lane and ST_ROWS rows; every 16-lane row of the wave (a 16 x 16 pixel block, two to
// three distinct labels) works on the smallest label still pending in it, so a wave needs two or three
// passes for four times the pixels of the former 16 x 4 blocks.  Per pass the partial sums of a lane (fixed-point limbs kept as integer-valued doubles, exact:
// |limb| * 256 pixels < 2^53 by the choice of the scales) go through one transposed DPP reduction over the
// 16 lanes, after which lane j of the row owns the total of quantity j and adds it to the workgroup's LDS
// slot of the label (open addressing; a full table falls back to global atomics).
// (RW: rows per lane -- ST_ROWS, or half of that for float64 sources (the filter responses of the texture path): 16 rows of three
// doubles in flight plus their IEEE divisions took 248 registers, two waves per SIMD)
// (one channel: the first pass keeps four limbs per row in registers -- eight rows; the second two -- sixteen)
template <typename T, int NC = 3, int PASS = 1> struct StatRows { static constexpr int value = (sizeof(T) == 8 || (NC == 1 && PASS == 1)) ? ST_ROWS / 2 : ST_ROWS; };

// NC = 1: one gray plane (the volumes, features_cython.pyx:144-219) -- one channel loaded and summed into the columns of channel 0,
// the five quantities of the first pass (count, two limbs of the value sum, two of the squared sum) through ONE transposed reduction
template <typename T, int PASS, bool U8INT, int NC = 3>
__global__ void __launch_bounds__(256, 4)      // (four workgroups a CU = four waves a SIMD: the allocator aims at <= 128 registers)
k_color_stats(const T *__restrict__ img, const int32_t *__restrict__ labels, StatParams sp,
              const float *__restrict__ mean32, long long *__restrict__ acc)
{
    constexpr int RW = StatRows<T, NC, PASS>::value;
    ZSHIFT(img, sp.zs); ZSHIFT(labels, sp.zs); ZSHIFT(mean32, sp.zs); ZSHIFT(acc, sp.zs); ZSHIFT(sp.ssq_dev, sp.zs);
    constexpr int NQ = (PASS == 1) ? 13 : 6;
    __shared__ int keys[ST_SLOTS];
    __shared__ long long lacc[ST_SLOTS][13];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid < ST_SLOTS) {
        keys[tid] = -1;
        for (int j = 0; j < NQ; ++j) lacc[tid][j] = 0;
    }
    __syncthreads();
    const int x = blockIdx.x * 64 + lane;
    const int y0 = (blockIdx.y * 4 + wave) * RW;
    // uint8 image, first pass: the terms are small integers (v <= 255, v * v <= 65025 -- the float32 product of the
    // reference is exact), so the block sums are plain int32 sums and the fixed-point limbs are (sum * 2^shift, 0):
    // the same accumulator contents as the general path at a fraction of the instructions
    static_assert(!U8INT || (PASS == 1 && sizeof(T) == 1 && NC == 3), "integer block sums: uint8 colour image, first pass");
    int lab[RW];
    float v[RW][NC];
    double mul = sp.mul, div = sp.div;
    bool dead = false;
    if (sp.prescale == 2) {                      // the L2 norm of the response stays on the device (imsegm_image2d_lm_features)
        const double norm = sqrt(*sp.ssq_dev);
        dead = !(norm > 0.0) || norm > DBL_MAX;
        mul = log(1.0 + norm) / 0.03;
        div = norm;
    }
#pragma unroll
    for (int r = 0; r < RW; ++r) {
        const int y = y0 + r;
        const bool ok = (y < sp.H) && (x < sp.W);
        const size_t p = ok ? (size_t)y * sp.W + x : 0;
        lab[r] = ok ? labels[p] : 0x7fffffff;
        if (NC == 3 && sizeof(T) == 1 && !sp.planar && !sp.prescale && p + 1 < sp.n_pixels) {
            // interleaved uint8: the three bytes of a pixel through one unaligned 32-bit load (the byte after them belongs
            // to the next pixel): a third of the load instructions and of the cache-line accesses of a wave
            uint32_t w;
            __builtin_memcpy(&w, reinterpret_cast<const uint8_t *>(img) + 3 * p, 4);
            v[r][0] = (float)(w & 0xffu);
            v[r][NC > 1 ? 1 : 0] = (float)((w >> 8) & 0xffu);
            v[r][NC > 1 ? 2 : 0] = (float)((w >> 16) & 0xffu);
            continue;
        }
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const size_t idx = NC == 1 ? p : (sp.planar ? (size_t)c * sp.plane_stride + p : 3 * p + c);
            v[r][c] = sp.prescale ? (dead ? 0.f : (float)(((double)img[idx] * mul) / div)) : load_f32(img, idx);
        }
    }
    double limb[(NC == 1 && PASS == 1) ? RW : 1][(NC == 1 && PASS == 1) ? 4 : 1];
    if constexpr (NC == 1 && PASS == 1) {
#pragma unroll
        for (int r = 0; r < RW; ++r) {
            const float val = v[r][0];
            const float term = __fmul_rn(val, val);
            const double t = (double)term * sp.scale_e, h = trunc(t);
            const double tv = (double)val * sp.scale_v, hv = trunc(tv);
            limb[r][0] = hv;
            limb[r][1] = trunc((tv - hv) * 4294967296.0);
            limb[r][2] = h;
            limb[r][3] = trunc((t - h) * 4294967296.0);
        }
    }
    while (true) {
        int mine = 0x7fffffff;
#pragma unroll
        for (int r = 0; r < RW; ++r) mine = min(mine, lab[r]);
        if (!__any(mine != 0x7fffffff)) break;
        const int k = row16_min_i32(mine);              // uniform over the 16-lane row; 0x7fffffff: row is done
        if constexpr (U8INT) {
            int qi[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };     // n, sums of v (3), sums of v * v (3)
#pragma unroll
            for (int r = 0; r < RW; ++r) {
                if (lab[r] != k || k == 0x7fffffff) continue;
                lab[r] = 0x7fffffff;
                qi[0] += 1;
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const int val = (int)v[r][c];
                    qi[1 + c] += val;
                    qi[4 + c] += val * val;
                }
            }
            const int toti = row16_reduce8_i32(qi, lane);
            const int j = ((lane >> 3) & 1) * 4 + ((lane >> 2) & 1) * 2 + ((lane >> 1) & 1);
            int slot = -1;
            if ((lane & 15) == 0 && k != 0x7fffffff) {
                int sidx = k & (ST_SLOTS - 1), probes = 0;
                while (probes < ST_SLOTS) {
                    int old = atomicCAS(&keys[sidx], -1, k);
                    if (old == -1 || old == k) break;
                    sidx = (sidx + 1) & (ST_SLOTS - 1);
                    ++probes;
                }
                slot = probes < ST_SLOTS ? sidx : -1;
            }
            slot = __shfl(slot, lane & 48, 64);
            if ((lane & 1) == 0 && j < 7 && toti != 0 && k != 0x7fffffff) {
                // accumulator columns: [0] count, [1 + 2c] high limb of the value sums, [7 + 2c] of the squared sums
                const int col = j == 0 ? 0 : (j < 4 ? 1 + 2 * (j - 1) : 7 + 2 * (j - 4));
                const long long tot = j == 0 ? (long long)toti
                                             : (long long)toti * (long long)(j < 4 ? sp.scale_v : sp.scale_e);
                if (slot >= 0) atomic_add_i64(&lacc[slot][col], tot);
                else atomic_add_i64(acc + (size_t)k * 13 + col, tot);
            }
            continue;
        }
        // LDS open-addressing slot of the row's label (found by the first lane of the row)
        int slot = -1;
        if ((lane & 15) == 0 && k != 0x7fffffff) {
            int sidx = k & (ST_SLOTS - 1), probes = 0;
            while (probes < ST_SLOTS) {
                int old = atomicCAS(&keys[sidx], -1, k);
                if (old == -1 || old == k) break;
                sidx = (sidx + 1) & (ST_SLOTS - 1);
                ++probes;
            }
            slot = probes < ST_SLOTS ? sidx : -1;
        }
        slot = __shfl(slot, lane & 48, 64);
        if constexpr (NC == 1) {
            // (one channel, eight rows per lane: the limbs of a voxel are formed ONCE, in front of the loop over the labels of the row
            // -- 4 x 8 doubles per lane --, and a pass over a label only adds the limbs of its voxels: 16 rows and the limbs formed
            // again for every label of a 16 x 16 block were 4.5 + 2.1 ms for the 2^30 voxels of config 5)
            const bool live = k != 0x7fffffff;
            const float m32 = (PASS == 2 && live) ? mean32[3 * k] : 0.f;
            double q[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) q[j] = 0;
#pragma unroll
            for (int r = 0; r < RW; ++r) {
                const bool on = lab[r] == k && live;
                lab[r] = on ? 0x7fffffff : lab[r];
                if (PASS == 1) {
                    q[0] += on ? limb[r][0] : 0.0;
                    q[1] += on ? limb[r][1] : 0.0;
                    q[2] += on ? limb[r][2] : 0.0;
                    q[3] += on ? limb[r][3] : 0.0;
                    q[6] += on ? 1.0 : 0.0;
                } else if (on) {
                    // (second pass: the mean is the label's -- one load per pass over a label, not one gather per voxel: forming
                    // these two limbs in front of the loop like the first pass's four measured 2.44 against 2.12 ms)
                    const float d = __fsub_rn(v[r][0], m32);
                    float term = __fmul_rn(d, d);
                    asm volatile("" : "+v"(term));
                    const double t = (double)term * sp.scale_e, h = trunc(t);
                    q[0] += h;
                    q[1] += trunc((t - h) * 4294967296.0);
                }
            }
            const long long tot = (long long)row16_reduce8_f64(reinterpret_cast<const double (&)[8]>(q), lane);
            const int j = ((lane >> 3) & 1) * 4 + ((lane >> 2) & 1) * 2 + ((lane >> 1) & 1);
            const bool owner = (lane & 1) == 0 && (PASS == 1 ? (j < 4 || j == 6) : j < 2);
            if (owner && tot != 0 && live) {
                // columns of channel 0: [0] count, [1..2] value sum, [7..8] squared / variance sum (LDS of the second pass: 0..1)
                const int col = PASS == 1 ? (j == 6 ? 0 : (j < 2 ? 1 + j : 5 + j)) : j;
                if (slot >= 0)
                    atomic_add_i64(&lacc[slot][col], tot);
                else
                    atomic_add_i64(acc + (size_t)k * 13 + ((PASS == 1) ? col : 7 + col), tot);
            }
            continue;
        }
        if constexpr (NC == 3) {
        // partial sums of this lane, eight at a time (one transposed reduction over the 16-lane row each: sixteen at once cost
        // 250+ registers): PASS 1 -> round 0: the value sums (two limbs per channel) and the count, round 1: the sums of v * v;
        // PASS 2 -> one round: the sums of (v - mean32)^2
        auto one_round = [&](auto round_tag) {
            constexpr int round = decltype(round_tag)::value;
            double q[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) q[j] = 0;
#pragma unroll
            for (int r = 0; r < RW; ++r) {
                if (lab[r] != k || k == 0x7fffffff) continue;
                if (PASS == 1 && round == 0) q[6] += 1.0;
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    float term;
                    if (PASS == 1) {
                        const float val = v[r][c];
                        term = round == 0 ? val : __fmul_rn(val, val);
                    } else {
                        const float d = __fsub_rn(v[r][c], mean32[3 * k + c]);
                        term = __fmul_rn(d, d);
                    }
                    // (the limbs are formed HERE, per label pass: hoisted out of the pass loop as loop invariants -- what the
                    // optimiser does when it can see through -- they are 4 x 3 x RW doubles, 400+ registers, one wave per SIMD)
                    asm volatile("" : "+v"(term));
                    const double t = (double)term * ((PASS == 1 && round == 0) ? sp.scale_v : sp.scale_e), h = trunc(t);
                    q[2 * c] += h;
                    q[2 * c + 1] += trunc((t - h) * 4294967296.0);
                }
            }
            const long long tot = (long long)row16_reduce8_f64(reinterpret_cast<const double (&)[8]>(q), lane);
            const int j = ((lane >> 3) & 1) * 4 + ((lane >> 2) & 1) * 2 + ((lane >> 1) & 1);
            const bool owner = (lane & 1) == 0 && (j < 6 || (PASS == 1 && round == 0 && j == 6));
            if (owner && tot != 0 && k != 0x7fffffff) {
                // accumulator columns: [0] count, [1..6] value sums, [7..12] squared / variance sums; the LDS slots of the second
                // pass hold columns 7..12 at 0..5
                const int col = PASS == 1 ? (round == 0 ? (j == 6 ? 0 : 1 + j) : 7 + j) : j;
                if (slot >= 0)
                    atomic_add_i64(&lacc[slot][col], tot);
                else
                    atomic_add_i64(acc + (size_t)k * 13 + ((PASS == 1) ? col : 7 + col), tot);
            }
        };
        one_round(std::integral_constant<int, 0>());
        if (PASS == 1) one_round(std::integral_constant<int, 1>());
        // (this label of the row is through)
#pragma unroll
        for (int r = 0; r < RW; ++r)
            if (lab[r] == k) lab[r] = 0x7fffffff;
        }
    }
    __syncthreads();
    for (int i = tid; i < ST_SLOTS * NQ; i += 256) {
        int slot = i / NQ, j = i - slot * NQ;
        int k = keys[slot];
        if (k < 0) continue;
        long long val = lacc[slot][j];
        if (val == 0) continue;
        int col = (PASS == 1) ? j : 7 + j;
        atomic_add_i64(acc + (size_t)k * 13 + col, val);
    }
}

__global__ void k_stats_clear(long long *acc, int K, int from, int to, size_t zs)
{
    ZSHIFT(acc, zs);
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= K) return;
    for (int j = from; j < to; ++j) acc[(size_t)i * 13 + j] = 0;
}

// normColorFeatures (features_cython.pyx:59-78): divide by the pixel count where count > 0
// (and the energy columns go back to zero for the variance pass, which sums into them again)
__global__ void k_stats_finalize1(long long *__restrict__ acc, StatParams sp, double *mean_out, double *energy_out,
                                  float *mean32)
{
    ZSHIFT(acc, sp.zs); ZSHIFT(mean_out, sp.zs); ZSHIFT(energy_out, sp.zs); ZSHIFT(mean32, sp.zs);
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= sp.K) return;
    long long *a = acc + (size_t)k * 13;
    long long n = a[0];
    for (int c = 0; c < 3; ++c) {
        double sv = (i64_to_double(a[1 + 2 * c]) + i64_to_double(a[2 + 2 * c]) * (1.0 / 4294967296.0)) / sp.scale_v;
        double se = (i64_to_double(a[7 + 2 * c]) + i64_to_double(a[8 + 2 * c]) * (1.0 / 4294967296.0)) / sp.scale_e;
        double m = n > 0 ? sv / (double)n : 0.0;
        double e = n > 0 ? se / (double)n : 0.0;
        if (mean_out) mean_out[3 * k + c] = m;
        if (energy_out) energy_out[3 * k + c] = e;
        mean32[3 * k + c] = (float)m;        // np.array(means, dtype=np.float32), descriptors.py:293
    }
    for (int j = 7; j < 13; ++j) a[j] = 0;
}

__global__ void k_stats_finalize2(const long long *__restrict__ acc, StatParams sp, double *var_out)
{
    ZSHIFT(acc, sp.zs); ZSHIFT(var_out, sp.zs);
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= sp.K) return;
    const long long *a = acc + (size_t)k * 13;
    long long n = a[0];
    for (int c = 0; c < 3; ++c) {
        double s = (i64_to_double(a[7 + 2 * c]) + i64_to_double(a[8 + 2 * c]) * (1.0 / 4294967296.0)) / sp.scale_e;
        var_out[3 * k + c] = n > 0 ? s / (double)n : 0.0;
    }
}

template <typename T>
static void launch_pass(int pass, const T *img, const int32_t *labels, StatParams sp, const float *mean32,
                        long long *acc, hipStream_t st, int nz)
{
    dim3 grid(cdiv(sp.W, 64), cdiv(sp.H, 4 * StatRows<T>::value), nz);
    if (sp.planar && sp.plane_stride == 0 && !sp.prescale) {
        grid.y = cdiv(sp.H, 4 * (pass == 1 ? StatRows<T, 1, 1>::value : StatRows<T, 1, 2>::value));         // one gray plane: channel 0 only (the others stay 0)
        if (pass == 1) hipLaunchKernelGGL((k_color_stats<T, 1, false, 1>), grid, 256, 0, st, img, labels, sp, mean32, acc);
        else hipLaunchKernelGGL((k_color_stats<T, 2, false, 1>), grid, 256, 0, st, img, labels, sp, mean32, acc);
        return;
    }
    if (pass == 1 && sizeof(T) == 1 && sp.u8_int)
        hipLaunchKernelGGL((k_color_stats<T, 1, sizeof(T) == 1>), grid, 256, 0, st, img, labels, sp, mean32, acc);
    else if (pass == 1)
        hipLaunchKernelGGL((k_color_stats<T, 1, false>), grid, 256, 0, st, img, labels, sp, mean32, acc);
    else
        hipLaunchKernelGGL((k_color_stats<T, 2, false>), grid, 256, 0, st, img, labels, sp, mean32, acc);
}

static double pow2_scale(double n_pixels, double maxabs)
{
    // largest power of two with n_pixels * maxabs * scale < 2^62 (int64 accumulators) and
    // 256 * maxabs * scale < 2^52 (exact fp64 sums inside one wave), capped at 2^30
    int e_n, e_m;
    frexp(n_pixels, &e_n);
    frexp(maxabs > 1.0 ? maxabs : 1.0, &e_m);
    int sh = 62 - e_n - e_m;
    if (sh > 44 - e_m) sh = 44 - e_m;
    if (sh > 30) sh = 30;
    return ldexp(1.0, sh);
}

int launch_color_stats(const void *img, int dtype, const int32_t *labels, int H, int W, int K, double maxabs,
                       int want_var, long long *acc, double *mean_out, double *energy_out, double *var_out,
                       float *mean32_scratch, hipStream_t st, int planar, int prescale, double mul, double div,
                       long plane_stride, const double *ssq_dev, ZBatch zb)
{
    StatParams sp;
    sp.zs = zb.zs;
    const unsigned nz = (unsigned)zb.nz;
    sp.H = H; sp.W = W; sp.K = K;
    sp.n_pixels = (size_t)H * W;
    sp.planar = planar; sp.prescale = prescale; sp.mul = mul; sp.div = div; sp.ssq_dev = ssq_dev;
    sp.plane_stride = plane_stride >= 0 ? (size_t)plane_stride : (size_t)H * W;
    sp.scale_v = pow2_scale((double)H * W, maxabs);
    sp.scale_e = pow2_scale((double)H * W, 4.0 * maxabs * maxabs);
    sp.u8_int = (dtype == DT_U8 && !prescale && sp.scale_v >= 1.0 && sp.scale_e >= 1.0) ? 1 : 0;
    const dim3 kgrid(cdiv(K, 256), 1, nz);
    hipLaunchKernelGGL(k_stats_clear, kgrid, 256, 0, st, acc, K, 0, 13, zb.zs);
    if (dtype == DT_U8) launch_pass<uint8_t>(1, (const uint8_t *)img, labels, sp, mean32_scratch, acc, st, zb.nz);
    else if (dtype == DT_F32) launch_pass<float>(1, (const float *)img, labels, sp, mean32_scratch, acc, st, zb.nz);
    else launch_pass<double>(1, (const double *)img, labels, sp, mean32_scratch, acc, st, zb.nz);
    hipLaunchKernelGGL(k_stats_finalize1, kgrid, 256, 0, st, acc, sp, mean_out, energy_out, mean32_scratch);
    if (want_var) {
        if (dtype == DT_U8) launch_pass<uint8_t>(2, (const uint8_t *)img, labels, sp, mean32_scratch, acc, st, zb.nz);
        else if (dtype == DT_F32) launch_pass<float>(2, (const float *)img, labels, sp, mean32_scratch, acc, st, zb.nz);
        else launch_pass<double>(2, (const double *)img, labels, sp, mean32_scratch, acc, st, zb.nz);
        hipLaunchKernelGGL(k_stats_finalize2, kgrid, 256, 0, st, acc, sp, var_out);
    }
    HIP_TRY(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------------------------
// Segmented label histogram: hist[label][annot] += 1 -- /root/reference/imsegm/labeling.py:208-247
// (`histogram_regions_labels_counts`, a per-pixel Python loop :244-245), used to label superpixels from an
// annotation (pipelines.py:284).  A lane owns LH_RUN consecutive pixels of the flat image (two 16-byte loads per
// array, a wave reads 2 KB contiguous) and merges runs of equal (label, annot) pairs in registers: neighbouring
// pixels nearly always agree, so the integer atomics (order independent, exact) drop to ~1 per lane.
// HBM bound: 8 B/pixel.
// ---------------------------------------------------------------------------------------------
constexpr int LH_RUN = 8;

__global__ void __launch_bounds__(256)
k_label_hist(const int32_t *__restrict__ labels, const int32_t *__restrict__ annot, size_t n, int K, int nb,
             unsigned long long *__restrict__ hist)
{
    const size_t base = ((size_t)blockIdx.x * 256 + threadIdx.x) * LH_RUN;
    if (base >= n) return;
    int lb[LH_RUN], an[LH_RUN];
    if (base + LH_RUN <= n) {
        const int4 l0 = *reinterpret_cast<const int4 *>(labels + base), l1 = *reinterpret_cast<const int4 *>(labels + base + 4);
        const int4 a0 = *reinterpret_cast<const int4 *>(annot + base), a1 = *reinterpret_cast<const int4 *>(annot + base + 4);
        lb[0] = l0.x; lb[1] = l0.y; lb[2] = l0.z; lb[3] = l0.w; lb[4] = l1.x; lb[5] = l1.y; lb[6] = l1.z; lb[7] = l1.w;
        an[0] = a0.x; an[1] = a0.y; an[2] = a0.z; an[3] = a0.w; an[4] = a1.x; an[5] = a1.y; an[6] = a1.z; an[7] = a1.w;
    } else {
#pragma unroll
        for (int i = 0; i < LH_RUN; ++i) {
            const bool ok = base + i < n;
            lb[i] = ok ? labels[base + i] : -1;
            an[i] = ok ? annot[base + i] : -1;
        }
    }
    long long cur = -1;            // flat bin of the current run, -1: not counted
    unsigned run = 0;
#pragma unroll
    for (int i = 0; i < LH_RUN; ++i) {
        const bool ok = lb[i] >= 0 && lb[i] < K && an[i] >= 0 && an[i] < nb;
        const long long bin = ok ? (long long)lb[i] * nb + an[i] : -1;
        if (bin != cur) {
            if (cur >= 0) atomicAdd(hist + cur, (unsigned long long)run);
            cur = bin;
            run = 0;
        }
        run++;
    }
    if (cur >= 0) atomicAdd(hist + cur, (unsigned long long)run);
}

int launch_label_hist(const int32_t *labels, const int32_t *annot, size_t n, int K, int nb, unsigned long long *hist,
                      hipStream_t st)
{
    HIP_TRY(hipMemsetAsync(hist, 0, (size_t)K * nb * sizeof(unsigned long long), st));
    if (n) hipLaunchKernelGGL(k_label_hist, cdiv((long)((n + LH_RUN - 1) / LH_RUN), 256), 256, 0, st, labels, annot, n, K, nb, hist);
    HIP_TRY(hipGetLastError());
    return 0;
}

}  // namespace imsegm
