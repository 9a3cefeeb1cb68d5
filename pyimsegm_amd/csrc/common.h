// common.h -- shared declarations of libimsegm_hip (gfx950 only).
//
// Device arithmetic helpers here define the bit-exact contract with the CPU oracle
// (oracle/imsegm_oracle.c): the library is compiled with -ffp-contract=off, so every expression
// below rounds exactly once per operation, as the gcc-compiled oracle does.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <type_traits>

#define IMSEGM_WAVE 64

namespace imsegm {

void set_error(const std::string &msg);
bool hip_ok(hipError_t e, const char *what, const char *file, int line);

#define HIP_TRY(expr)                                                                   \
    do {                                                                                \
        if (!::imsegm::hip_ok((expr), #expr, __FILE__, __LINE__)) return -1;            \
    } while (0)

// Debug / experiment switches of the library.  The process environment is read ONCE (first use) and again only when
// imsegm_debug_reload_env() is called (tests flip switches at run time): nothing on the per-image path looks at the environment.
struct Knobs {
    bool slic_graph, pre_3pass, separate_finalize, fuse_finalize, conn_general, gc_no_topo_regs,
        adjacency_table,       // IMSEGM_ADJACENCY_TABLE: the neighbour table of a label volume's graph for every K (default: beyond 46 000 labels)
        cc_merge_full,         // IMSEGM_CC_MERGE_FULL: the voxel-by-voxel merge passes of rounds 2 - 4 -- measure.label with thirteen unions per voxel, connectivity with
                               // its per-voxel loads -- instead of the row-segment kernels (tests, A/B)
        terms_one_workgroup,   // IMSEGM_TERMS_ONE_WORKGROUP: the graph-cut terms of a volume by the one workgroup that computes an image's (tests, A/B)
        label_general,         // IMSEGM_LABEL_GENERAL: measure.label of a volume always by union-find, also where the label map is known to come from
                               // the connectivity pass (tests, A/B: connectivity.hip launch_label_connected)
        sep_wide_tile;         // IMSEGM_SEP_WIDE_TILE: the separable kernels of side 33 on the 64 x 16 tile of round 4's first half (tests, A/B)
    int brick_cap;             // IMSEGM_BRICK_CAP (0: default)
    int gc_lds_level;          // IMSEGM_GC_LDS_LEVEL (default 4)
    int gc_threads;            // IMSEGM_GC_THREADS (0: default)
    int gc_grid_min_sites;     // IMSEGM_GC_GRID_MIN_SITES (0: default 8192): graphs of that many sites that do not fit one CU's LDS go to the grid-wide kernel
    int gc_grid_blocks;        // IMSEGM_GC_GRID_BLOCKS (0: one per CU)
    bool gc_one_workgroup;     // IMSEGM_GC_ONE_WORKGROUP: never the grid-wide kernel (tests, A/B)
    bool gc_grid_test_absent;  // IMSEGM_GC_GRID_TEST_ABSENT: one workgroup of the grid-wide kernel never arrives (test of the bounded wait + fall-back)
    int fused_bitmap_mb;       // IMSEGM_FUSED_BITMAP_MB (0: what the device has free): ceiling of the fused call's two K x K bit arrays (tests of the fall-back)
    std::string phase_dump;    // IMSEGM_PHASE_DUMP (file name, empty: none)
};
const Knobs &knobs();
void reload_knobs();

constexpr int IMSEGM_MAX_DEVICES = 64;      // size of the per-device caches of function attributes
__host__ __device__ static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// ---------------------------------------------------------------------------------------------
// several images of one size in ONE launch (csrc/batch.hip): image b is blockIdx.z and works on the same buffers `zs` bytes
// further on per image -- every per-image buffer of a batch lives at the same offset of its image's slice of one arena, so ONE
// stride shifts them all.  A single image is the batch of one (gridDim.z = 1, shift 0).
// ---------------------------------------------------------------------------------------------
struct ZBatch {
    int nz = 1;        // images per launch (gridDim.z)
    size_t zs = 0;     // bytes from a buffer of image b to the same buffer of image b + 1
};
template <typename T> __device__ __forceinline__ T *zshift(T *p, size_t zs)
{
    // Byte arithmetic ON THE POINTER (never through an integer): the compiler knows a kernel argument to point to global memory
    // only as long as it can follow the pointer -- one uintptr_t round trip and every access behind it is a FLAT instruction
    // (measured: the 2048^2 line fell from 6.7 to 4.5 Gpixel/s, 43 -> 1 797 flat instructions in slic.hip).
    // Null stays null for EVERY image of a batch: optional arguments ("output not wanted", `order` of k_small_bfs_wave, `classes`
    // of k_label_lut, the scaler vectors of TermsArgs) do arrive together with a stride, and the kernels test them AFTER the
    // shift -- the select below is a scalar instruction per pointer and keeps both the provenance and the address space
    // (relying on the compiler to fold `p + b * zs == nullptr` back onto `p` was undefined behaviour).
    typedef typename std::conditional<std::is_const<T>::value, const char, char>::type byte_t;
    T *const shifted = reinterpret_cast<T *>(reinterpret_cast<byte_t *>(p) + (size_t)blockIdx.z * zs);
    return p ? shifted : nullptr;
}
// the same for a pointer that is NEVER null by construction (carved from a session arena): no select -- the hot SLIC kernels shift
// two dozen of them per wave on a scalar unit that four SIMDs share
template <typename T> __device__ __forceinline__ T *zshift_nn(T *p, size_t zs)
{
    typedef typename std::conditional<std::is_const<T>::value, const char, char>::type byte_t;
    return reinterpret_cast<T *>(reinterpret_cast<byte_t *>(p) + (size_t)blockIdx.z * zs);
}
#define ZSHIFT(p, zs) (p) = ::imsegm::zshift((p), (zs))
#define ZSHIFT_NN(p, zs) (p) = ::imsegm::zshift_nn((p), (zs))

// ---------------------------------------------------------------------------------------------
// deterministic elementary functions (mirror of oracle det_rcbrt / orc_det_cbrt / orc_det_pow24)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ double det_cbrt(double x)
{
    long long i = __double_as_longlong(x);
    i = 0x553ef0ff289dd796LL - i / 3;
    double y = __longlong_as_double(i);
    const double third = 1.0 / 3.0;
#pragma unroll
    for (int it = 0; it < 5; ++it) {
        double y3 = y * y * y;
        double r = 1.0 - x * y3;
        y = y + y * (r * third);
    }
    double yy = y * y;
    double c = x * yy;
    double e = c * c * c - x;
    c = c - e * (yy * third);
    return c;
}

// x / C for a constant C, correctly rounded, in three instructions instead of the ~14 of an IEEE division (Markstein: with
// y = RN(1 / C), q = RN(x * y) and r = x - q * C formed exactly by an FMA, RN(q + r * y) = RN(x / C) whenever the significand of C
// is not all ones -- and as long as r does not underflow: for x = 0 or 2^-500 <= |x| <= 2^500 only, which the caller guarantees
// (the XYZ values of a uint8 image: 0 or >= 5e-5).  Checked against the division on 4 * 10^8 random arguments per constant
// (0.95047, 1.08883); same bits as the oracle's plain division.
__device__ __forceinline__ double div_by_const_in_range(double x, double c, double rc)
{
    const double q = x * rc;
    const double r = fma(-q, c, x);
    return fma(r, rc, q);
}
#define DIV_CONST_IN_RANGE(x, C) ::imsegm::div_by_const_in_range((x), (C), 1.0 / (C))

__device__ __forceinline__ double det_pow24(double t)
{
    long long i = __double_as_longlong(t);
    i = 0x4cb8a8c154c985f0LL - i / 5;
    double y = __longlong_as_double(i);
#pragma unroll
    for (int it = 0; it < 5; ++it) {
        double y2 = y * y;
        double y5 = y2 * y2 * y;
        double r = 1.0 - t * y5;
        y = y + y * (r * 0.2);
    }
    double p = t * y;
    return p * p * p;
}

// skimage.color.rgb2lab (colorconv.py, D65 / 2 deg), one pixel in [0, 1]
__device__ __forceinline__ void rgb2lab_px(double r, double g, double b, double &L, double &A, double &B)
{
    double lin[3] = { r, g, b };
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        double v = lin[c];
        lin[c] = (v > 0.04045) ? det_pow24((v + 0.055) / 1.055) : v / 12.92;
    }
    double X = lin[0] * 0.412453 + lin[1] * 0.357580 + lin[2] * 0.180423;
    double Y = lin[0] * 0.212671 + lin[1] * 0.715160 + lin[2] * 0.072169;
    double Z = lin[0] * 0.019334 + lin[1] * 0.119193 + lin[2] * 0.950227;
    double f[3] = { X / 0.95047, Y / 1.0, Z / 1.08883 };
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        double t = f[c];
        f[c] = (t > 0.008856) ? det_cbrt(t) : 7.787 * t + 16.0 / 116.0;
    }
    L = (116.0 * f[1]) - 16.0;
    A = 500.0 * (f[0] - f[1]);
    B = 200.0 * (f[1] - f[2]);
}

// ---------------------------------------------------------------------------------------------
// exact order-independent fixed-point accumulation
// ---------------------------------------------------------------------------------------------
__host__ __device__ __forceinline__ double i64_to_double(long long v)
{
    /* (double)(int32 high) * 2^32 + (double)(uint32 low): one rounding, round-to-nearest-even */
    int h = (int)(v >> 32);
    unsigned int l = (unsigned int)(v & 0xffffffffLL);
    return (double)h * 4294967296.0 + (double)l;
}

// Fixed-point format of the centroid colour sums (oracle: orc_fix_bits): t = trunc(v * 2^f), f = 46 - e with
// 2^e > max |v|.  Sums of up to 64 such integers are exact in fp64 (< 2^52), sums of a workgroup in int64; the
// global sum lives in two int64 limbs, sum = hi * 2^24 + lo, which fix_value() brings to the canonical split
// (0 <= lo < 2^24) before the one rounding to fp64 -- so the value is independent of how partial sums were split.
__device__ __forceinline__ int fix_bits_of(double maxabs)
{
    int e = 1;
    if (maxabs > 0.0) (void)frexp(maxabs, &e);
    return 46 - e;
}
__host__ __device__ __forceinline__ double fix_value(long long hi, long long lo, double finv)
{
    hi += lo >> 24;
    lo &= 0xffffff;
    return (i64_to_double(hi) * 16777216.0 + (double)lo) * finv;
}

// ---------------------------------------------------------------------------------------------
// wave64 helpers
// ---------------------------------------------------------------------------------------------
// lane i receives the value of lane i - 1 (lane 0 keeps `first`) / of lane i + 1 (lane 63 keeps `last`): ONE DPP move across the
// whole wave (wave_shr:1 / wave_shl:1), no LDS round trip as __shfl_up / __shfl_down take
__device__ __forceinline__ int lane_prev(int v, int first) { return __builtin_amdgcn_update_dpp(first, v, 0x138, 0xf, 0xf, false); }
__device__ __forceinline__ int lane_next(int v, int last) { return __builtin_amdgcn_update_dpp(last, v, 0x130, 0xf, 0xf, false); }

// Union-find with the smaller index as the root (atomicMin on the parent of the larger root), TWO unions of a lane side by side:
// (a0, b0) and, where a1 >= 0, (a1, b1).  The four walks to the roots advance together -- four loads in flight per step instead
// of the one of a find after the other -- and a union whose atomicMin lost a race goes on from what it saw.  Unions of one lane
// may touch the same sets: atomicMin keeps every interleaving a forest whose roots are the smallest indices.
__device__ __forceinline__ void union2_min_root(int32_t *parent, int a0, int b0, int a1, int b1)
{
    bool on0 = a0 >= 0, on1 = a1 >= 0;
    while (on0 || on1) {
        const int pa0 = on0 ? parent[a0] : 0, pb0 = on0 ? parent[b0] : 0;
        const int pa1 = on1 ? parent[a1] : 0, pb1 = on1 ? parent[b1] : 0;
        if (on0) {
            if (pa0 == a0 && pb0 == b0) {
                const int hi = max(a0, b0), lo = min(a0, b0);
                const int old = hi == lo ? hi : atomicMin(&parent[hi], lo);
                on0 = old != hi;
                a0 = old;
                b0 = lo;
            } else {
                a0 = pa0;
                b0 = pb0;
            }
        }
        if (on1) {
            if (pa1 == a1 && pb1 == b1) {
                const int hi = max(a1, b1), lo = min(a1, b1);
                const int old = hi == lo ? hi : atomicMin(&parent[hi], lo);
                on1 = old != hi;
                a1 = old;
                b1 = lo;
            } else {
                a1 = pa1;
                b1 = pb1;
            }
        }
    }
}

__device__ __forceinline__ long long wave_sum_i64(long long v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
__device__ __forceinline__ int wave_sum_i32(int v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
__device__ __forceinline__ double wave_min_f64(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmin(v, __shfl_xor(v, off, 64));
    return v;
}
__device__ __forceinline__ double wave_max_f64(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_xor(v, off, 64));
    return v;
}

// Transposed multi-value reductions: sum M int64 values over the 64 lanes with M/2 + M/4 + ... + 1
// (+ log2(64/M) butterfly) exchanges instead of 6 * M.  On return every lane holds the complete
// total of ONE of the values: value index = (lane >> 3) for M = 8, (lane >> 2) for M = 16, so
// the M lanes {0, 64/M, 2*64/M, ...} can issue the M atomics in parallel.
__device__ __forceinline__ long long wave_reduce8_i64(const long long (&v)[8])
{
    const int lane = threadIdx.x & 63;
    bool up = lane & 32;
    long long a[4], b[2];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        long long send = up ? v[j] : v[j + 4], keep = up ? v[j + 4] : v[j];
        a[j] = keep + __shfl_xor(send, 32, 64);
    }
    up = lane & 16;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        long long send = up ? a[j] : a[j + 2], keep = up ? a[j + 2] : a[j];
        b[j] = keep + __shfl_xor(send, 16, 64);
    }
    up = lane & 8;
    long long send = up ? b[0] : b[1], keep = up ? b[1] : b[0];
    long long c = keep + __shfl_xor(send, 8, 64);
    c += __shfl_xor(c, 4, 64);
    c += __shfl_xor(c, 2, 64);
    c += __shfl_xor(c, 1, 64);
    return c;
}

// same exchange pattern for 8 values of which v[0..5] carry fp64 bit patterns (added as doubles,
// exact while the sums stay below 2^53) and v[6..7] are int64
__device__ __forceinline__ long long wave_reduce8_mixed(const long long (&v)[8])
{
    const int lane = threadIdx.x & 63;
    auto addm = [](long long a, long long b, bool is_f64) -> long long {
        return is_f64 ? __double_as_longlong(__longlong_as_double(a) + __longlong_as_double(b)) : a + b;
    };
    bool up = lane & 32;
    long long a[4], b[2];
    // after step 1 lanes with bit 5 set hold indices 4..7 (4, 5 fp64; 6, 7 int), the others 0..3 (fp64)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        long long send = up ? v[j] : v[j + 4], keep = up ? v[j + 4] : v[j];
        long long recv = __shfl_xor(send, 32, 64);
        a[j] = addm(keep, recv, !(up && j >= 2));
    }
    bool up2 = lane & 16;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        long long send = up2 ? a[j] : a[j + 2], keep = up2 ? a[j + 2] : a[j];
        long long recv = __shfl_xor(send, 16, 64);
        b[j] = addm(keep, recv, !(up && up2));          // indices 6, 7 live where bits 5 and 4 are set
    }
    bool up3 = lane & 8;
    const bool is_f64 = !(up && up2);
    long long send = up3 ? b[0] : b[1], keep = up3 ? b[1] : b[0];
    long long c = addm(keep, __shfl_xor(send, 8, 64), is_f64);
    c = addm(c, __shfl_xor(c, 4, 64), is_f64);
    c = addm(c, __shfl_xor(c, 2, 64), is_f64);
    c = addm(c, __shfl_xor(c, 1, 64), is_f64);
    return c;
}

__device__ __forceinline__ long long wave_reduce16_i64(const long long (&v)[16])
{
    const int lane = threadIdx.x & 63;
    bool up = lane & 32;
    long long a[8], b[4], c[2];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        long long send = up ? v[j] : v[j + 8], keep = up ? v[j + 8] : v[j];
        a[j] = keep + __shfl_xor(send, 32, 64);
    }
    up = lane & 16;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        long long send = up ? a[j] : a[j + 4], keep = up ? a[j + 4] : a[j];
        b[j] = keep + __shfl_xor(send, 16, 64);
    }
    up = lane & 8;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        long long send = up ? b[j] : b[j + 2], keep = up ? b[j + 2] : b[j];
        c[j] = keep + __shfl_xor(send, 8, 64);
    }
    up = lane & 4;
    long long send = up ? c[0] : c[1], keep = up ? c[1] : c[0];
    long long d = keep + __shfl_xor(send, 4, 64);
    d += __shfl_xor(d, 2, 64);
    d += __shfl_xor(d, 1, 64);
    return d;
}

// fp64 variants of the transposed reductions (used with integer-valued doubles, i.e. exact sums)
__device__ __forceinline__ double wave_reduce8_f64(const double (&v)[8])
{
    const int lane = threadIdx.x & 63;
    bool up = lane & 32;
    double a[4], b[2];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        double send = up ? v[j] : v[j + 4], keep = up ? v[j + 4] : v[j];
        a[j] = keep + __shfl_xor(send, 32, 64);
    }
    up = lane & 16;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        double send = up ? a[j] : a[j + 2], keep = up ? a[j + 2] : a[j];
        b[j] = keep + __shfl_xor(send, 16, 64);
    }
    up = lane & 8;
    double send = up ? b[0] : b[1], keep = up ? b[1] : b[0];
    double c = keep + __shfl_xor(send, 8, 64);
    c += __shfl_xor(c, 4, 64);
    c += __shfl_xor(c, 2, 64);
    c += __shfl_xor(c, 1, 64);
    return c;
}

__device__ __forceinline__ double wave_reduce16_f64(const double (&v)[16])
{
    const int lane = threadIdx.x & 63;
    bool up = lane & 32;
    double a[8], b[4], c[2];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        double send = up ? v[j] : v[j + 8], keep = up ? v[j + 8] : v[j];
        a[j] = keep + __shfl_xor(send, 32, 64);
    }
    up = lane & 16;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        double send = up ? a[j] : a[j + 4], keep = up ? a[j + 4] : a[j];
        b[j] = keep + __shfl_xor(send, 16, 64);
    }
    up = lane & 8;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        double send = up ? b[j] : b[j + 2], keep = up ? b[j + 2] : b[j];
        c[j] = keep + __shfl_xor(send, 8, 64);
    }
    up = lane & 4;
    double send = up ? c[0] : c[1], keep = up ? c[1] : c[0];
    double d = keep + __shfl_xor(send, 4, 64);
    d += __shfl_xor(d, 2, 64);
    d += __shfl_xor(d, 1, 64);
    return d;
}

// sum eight doubles per lane over each 16-lane row: 4 + 2 + 1 transposed exchanges + 1 plain (DPP
// row_mirror / row_half_mirror / quad_perm; no LDS traffic).  Returns, in every lane, the row total of the
// value with index ((lane >> 3) & 1) * 4 + ((lane >> 2) & 1) * 2 + ((lane >> 1) & 1).
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v)
{
    const long long b = __double_as_longlong(v);
#ifndef IMSEGM_DPP_PRESET
#define IMSEGM_DPP_PRESET 0
#endif
#if IMSEGM_DPP_PRESET
    const int lo = __builtin_amdgcn_update_dpp(0, (int)b, CTRL, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, 0xf, 0xf, false);
#else
    // (mirror / quad permutations read a valid lane everywhere: no preset of the destination, one instruction per half)
    const int lo = __builtin_amdgcn_mov_dpp((int)b, CTRL, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_mov_dpp((int)(b >> 32), CTRL, 0xf, 0xf, false);
#endif
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
__device__ __forceinline__ double row16_reduce8_f64(const double (&v)[8], int lane)
{
    double a[4], b[2];
    bool up = lane & 8;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const double send = up ? v[j] : v[j + 4], keep = up ? v[j + 4] : v[j];
        a[j] = keep + dpp_f64<0x140>(send);                 // row_mirror: lane i <-> 15 - i
    }
    up = lane & 4;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const double send = up ? a[j] : a[j + 2], keep = up ? a[j + 2] : a[j];
        b[j] = keep + dpp_f64<0x141>(send);                 // row_half_mirror: i <-> 7 - i
    }
    up = lane & 2;
    const double send = up ? b[0] : b[1], keep = up ? b[1] : b[0];
    double c = keep + dpp_f64<0x1b>(send);                  // quad_perm [3,2,1,0]: i <-> 3 - i
    c += dpp_f64<0xb1>(c);                                  // quad_perm [1,0,3,2]: i <-> i ^ 1
    return c;
}

// the same for eight ints per lane (lane pair j of the row holds the row total of value j; j as in row16_reduce8_f64)
__device__ __forceinline__ int row16_reduce8_i32(const int (&v)[8], int lane)
{
    int a[4], b[2];
    bool up = lane & 8;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int send = up ? v[j] : v[j + 4], keep = up ? v[j + 4] : v[j];
        a[j] = keep + __builtin_amdgcn_update_dpp(0, send, 0x140, 0xf, 0xf, false);
    }
    up = lane & 4;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int send = up ? a[j] : a[j + 2], keep = up ? a[j + 2] : a[j];
        b[j] = keep + __builtin_amdgcn_update_dpp(0, send, 0x141, 0xf, 0xf, false);
    }
    up = lane & 2;
    const int send = up ? b[0] : b[1], keep = up ? b[1] : b[0];
    int c = keep + __builtin_amdgcn_update_dpp(0, send, 0x1b, 0xf, 0xf, false);
    c += __builtin_amdgcn_update_dpp(0, c, 0xb1, 0xf, 0xf, false);
    return c;
}

// sixteen doubles per lane -> every lane of the 16-lane row ends up with the row total of value (lane & 15)
__device__ __forceinline__ double row16_reduce16_f64(const double (&v)[16], int lane)
{
    double a[8], b[4], c[2];
    bool up = lane & 8;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const double send = up ? v[j] : v[j + 8], keep = up ? v[j + 8] : v[j];
        a[j] = keep + dpp_f64<0x140>(send);                 // row_mirror
    }
    up = lane & 4;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const double send = up ? a[j] : a[j + 4], keep = up ? a[j + 4] : a[j];
        b[j] = keep + dpp_f64<0x141>(send);                 // row_half_mirror
    }
    up = lane & 2;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const double send = up ? b[j] : b[j + 2], keep = up ? b[j + 2] : b[j];
        c[j] = keep + dpp_f64<0x1b>(send);                  // quad_perm [3,2,1,0]
    }
    up = lane & 1;
    const double send = up ? c[0] : c[1], keep = up ? c[1] : c[0];
    return keep + dpp_f64<0xb1>(send);                      // quad_perm [1,0,3,2]
}

// minimum / maximum over the wave of a float that is never negative (+inf allowed): such floats order like their bit patterns, so
// the reduction is six integer min / max with DPP operands (row_shr 1 / 2 / 4 / 8, row_bcast 15 / 31 -- no LDS round trip, where
// __shfl_xor costs one ds_bpermute and its latency per step) and one v_readlane of lane 63; the result is wave uniform.
__device__ __forceinline__ float wave_min_nonneg_f32(float v)
{
    // as a maximum of (+inf bits - x) >= 0: a lane without a source then contributes 0 through bound_ctrl and the compiler folds the
    // DPP operand into v_max_i32 (with +inf as the value to keep it needs a preset and a separate move per step)
    int x = 0x7f800000 - __float_as_int(v);
    x = max(x, __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, false));
    x = max(x, __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, false));
    x = max(x, __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, false));
    x = max(x, __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, false));
    x = max(x, __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false));
    x = max(x, __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false));
    return __int_as_float(0x7f800000 - __builtin_amdgcn_readlane(x, 63));
}
__device__ __forceinline__ float wave_max_nonneg_f32(float v)
{
    int x = __float_as_int(v);
    x = max(x, __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, false));
    x = max(x, __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, false));
    x = max(x, __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, false));
    x = max(x, __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, false));
    x = max(x, __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false));
    x = max(x, __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false));
    return __int_as_float(__builtin_amdgcn_readlane(x, 63));
}

// minimum of an int over each 16-lane row (all lanes get it)
__device__ __forceinline__ int row16_min_i32(int v)
{
    v = min(v, __builtin_amdgcn_update_dpp(0, v, 0x140, 0xf, 0xf, false));
    v = min(v, __builtin_amdgcn_update_dpp(0, v, 0x141, 0xf, 0xf, false));
    v = min(v, __builtin_amdgcn_update_dpp(0, v, 0x1b, 0xf, 0xf, false));
    v = min(v, __builtin_amdgcn_update_dpp(0, v, 0xb1, 0xf, 0xf, false));
    return v;
}

__device__ __forceinline__ void atomic_add_i64(long long *p, long long v)
{
    atomicAdd(reinterpret_cast<unsigned long long *>(p), (unsigned long long)v);
}
// add the exact integer t to a two-limb global sum (hi at p[0], lo at p[1])
__device__ __forceinline__ void fix_add_global(long long *p, long long t)
{
    atomic_add_i64(p, t >> 24);
    atomic_add_i64(p + 1, t & 0xffffff);
}

}  // namespace imsegm
