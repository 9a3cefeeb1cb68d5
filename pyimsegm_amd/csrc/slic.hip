// slic.hip -- SLIC superpixel kernels for gfx950 (wave64, LDS-staged centroid tiles).
//
// Replaces the native boundary `skimage.segmentation.slic` called at
// /root/reference/imsegm/superpixels.py:61-63 (2D colour).  Arithmetic contract: bit-identical
// to oracle/imsegm_oracle.c (orc_slic_preprocess_color2d, orc_slic_iterate).
//
// Data layout in HBM: Lab image as three fp64 planes [3][H][W] (one 8-byte element per lane ->
// fully coalesced 512-byte wave loads); labels int32 [H][W]; centroid table SoA (cy, cx, cL, ca,
// cb as fp64[K], integer search windows int4[K]); accumulators int64 [K][9].
#include "slic.h"
#include <hip/hip_ext.h>
#include <atomic>
#include <cstdio>

namespace imsegm {

// ---------------------------------------------------------------------------------------------
// centroid table
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int4 search_window(double cy, double cx, int step_y, int step_x, int H, int W)
{
    // _slic.pyx: y_min = <Py_ssize_t>max(cy - 2 * step_y, 0); y_max = <Py_ssize_t>min(cy + 2 * step_y + 1, height)
    double a;
    int4 w;
    a = cy - (double)(2 * step_y);
    w.x = (int)(a > 0 ? a : 0.0);
    a = cy + (double)(2 * step_y);
    a = a + 1.0;
    w.y = (int)(a < (double)H ? a : (double)H);
    a = cx - (double)(2 * step_x);
    w.z = (int)(a > 0 ? a : 0.0);
    a = cx + (double)(2 * step_x);
    a = a + 1.0;
    w.w = (int)(a < (double)W ? a : (double)W);
    return w;
}

// several images per launch (ZBatch): every device pointer of the state moves to image blockIdx.z (fail_host is a host word
// shared by the batch; phase_prof is a profiling aid of single images)
__device__ __forceinline__ void zshift_state(SlicState &s)
{
    // every table of the state is carved from the session's arenas (api.hip slic_place_state): never null -- but for `done`, which
    // the launcher clears when the centroid update runs as separate finalize launches
    const size_t zs = s.zs;
    ZSHIFT_NN(s.cy, zs); ZSHIFT_NN(s.cx, zs); ZSHIFT_NN(s.cL, zs); ZSHIFT_NN(s.ca, zs); ZSHIFT_NN(s.cb, zs);
    ZSHIFT_NN(s.win, zs); ZSHIFT_NN(s.acc, zs); ZSHIFT_NN(s.premax, zs);
    ZSHIFT_NN(s.tile_cands, zs); ZSHIFT_NN(s.tile_count, zs); ZSHIFT_NN(s.tile_rec, zs); ZSHIFT_NN(s.tile_info, zs); ZSHIFT_NN(s.tile_k, zs); ZSHIFT_NN(s.tile_rows, zs);
    ZSHIFT_NN(s.leftover, zs); ZSHIFT_NN(s.leftover_count, zs); ZSHIFT_NN(s.mdc, zs); ZSHIFT_NN(s.drift, zs); ZSHIFT(s.done, zs);
}

// regular grid of skimage.util.regular_grid: centroid k = (iy, ix) in row-major order
__global__ void k_centroid_init(SlicState s, const double *__restrict__ init_yx)
{
    zshift_state(s);
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < SLIC_DRIFT_SLOTS) s.drift[k] = 0;
    if (k >= s.K) return;
    (void)init_yx;
    const int iy = k / s.grid_nx, ix = k - iy * s.grid_nx;
    double cy = (double)(s.grid_y0 + iy * s.grid_dy), cx = (double)(s.grid_x0 + ix * s.grid_dx);
    s.cy[k] = cy;
    s.cx[k] = cx;
    s.cL[k] = 0.0;
    s.ca[k] = 0.0;
    s.cb[k] = 0.0;
    s.win[k] = search_window(cy, cx, s.step_y, s.step_x, s.H, s.W);
    s.mdc[k] = 1.0;
    for (int j = 0; j < 9; ++j) s.acc[(size_t)k * 9 + j] = 0;
    if (s.done) s.done[k] = 0;
    if (k == 0) *s.leftover_count = 0;
}

// divide the accumulated sums, recompute the integer search windows, clear the accumulators.
// A centroid that received no pixel is dead from now on (NaN position in skimage): empty window.
// `drift_slot`: where the largest distance (per axis, rounded up) of a centroid from its grid node goes; the next
// k_slic_bin only scans the grid nodes that can reach its tile (any upper bound is valid there).
__device__ __forceinline__ void centroid_finalize_one(const SlicState &s, int k, int drift_slot)
{
    if (k == 0) *s.leftover_count = 0;
    if (k >= s.K) return;
    long long *a = s.acc + (size_t)k * 9;
    long long n = a[0];
    if (n == 0) {   // no pixel carries label k any more: it can never be assigned again
        s.win[k] = make_int4(0, 0, 0, 0);
    } else {
        double nn = (double)n;
        double cy = i64_to_double(a[1]) / nn;
        double cx = i64_to_double(a[2]) / nn;
        s.cy[k] = cy;
        s.cx[k] = cx;
        const double finv = ldexp(1.0, -fix_bits_of(*s.premax));
        s.cL[k] = fix_value(a[3], a[4], finv) / nn;
        s.ca[k] = fix_value(a[5], a[6], finv) / nn;
        s.cb[k] = fix_value(a[7], a[8], finv) / nn;
        s.win[k] = search_window(cy, cx, s.step_y, s.step_x, s.H, s.W);
        const int iy = k / s.grid_nx, ix = k - iy * s.grid_nx;
        const double dy = fabs(cy - (double)(s.grid_y0 + iy * s.grid_dy)), dx = fabs(cx - (double)(s.grid_x0 + ix * s.grid_dx));
        const int d = (int)ceil(fmax(dy, dx));
        if (d > 0) atomicMax(&s.drift[drift_slot], d);
    }
#pragma unroll
    for (int j = 0; j < 9; ++j) a[j] = 0;
}

__global__ void k_centroid_finalize(SlicState s, int drift_slot)
{
    zshift_state(s);
    centroid_finalize_one(s, blockIdx.x * blockDim.x + threadIdx.x, drift_slot);
}

constexpr int TILE_X = SLIC_TILE_X;   // one pixel column per lane
constexpr int TILE_Y = SLIC_TILE_Y;   // rows of a candidate-bin tile
constexpr int MAXC = SLIC_MAXC;

// ---- per-tile candidate lists ----------------------------------------------------------------
// One wave per 64 x 32 pixel tile collects the centroids whose search window intersects the tile,
// sorts them nearest-first and writes them as 96-byte records.  The assignment kernel then walks
// its list with wave-uniform (scalar) loads: no LDS staging, no barriers in front of the hot loop.
constexpr int BIN_TILES_PER_BLOCK = 4;     // one wave per tile, four tiles share one LDS copy of the windows
constexpr int BIN_MAX_K_LDS = 4096;        // centroids whose windows fit the LDS copy (64 KB)
constexpr int BIN_LOCAL_MAX = 512;         // grid nodes a tile may have to look at on the local path

__global__ void __launch_bounds__(256)
k_slic_bin(SlicState s, int tiles_x, int n_tiles, int max_cand, Cand *__restrict__ tile_cands,
           int *__restrict__ tile_count, int drift_slot)
{
    zshift_state(s);
    ZSHIFT(tile_cands, s.zs); ZSHIFT(tile_count, s.zs);
    extern __shared__ int4 lds_win[];                 // [min(K, BIN_MAX_K_LDS)]
    __shared__ int ck[BIN_TILES_PER_BLOCK][MAXC];
    __shared__ float ckey[BIN_TILES_PER_BLOCK][MAXC];
    __shared__ float clb[BIN_TILES_PER_BLOCK][MAXC];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int klds = min(s.K, BIN_MAX_K_LDS);
    // Centroids stay near their grid nodes: with `drift` = the largest displacement so far (k_centroid_finalize), only
    // the nodes within reach of the tile have to be looked at -- one or two wave iterations instead of K / 64.  The
    // exact test on the integer windows is the same either way; when the centroids have wandered far, scan them all.
    const int drift = s.drift[drift_slot];
    const int reach_y = 2 * s.step_y + 1 + drift, reach_x = 2 * s.step_x + 1 + drift;
    const bool local = s.grid_dy > 0 && s.grid_dx > 0 && (long)(2 * reach_y + TILE_Y + 2 * s.grid_dy) * (2 * reach_x + TILE_X + 2 * s.grid_dx) <=
                       (long)BIN_LOCAL_MAX * s.grid_dy * s.grid_dx;
    if (!local) {
        for (int k = threadIdx.x; k < klds; k += 256) lds_win[k] = s.win[k];
        __syncthreads();
    }
    const int tile = blockIdx.x * BIN_TILES_PER_BLOCK + wave;
    if (tile >= n_tiles) return;
    const int tx0 = (tile % tiles_x) * TILE_X, ty0 = (tile / tiles_x) * TILE_Y;
    const int tx1 = min(tx0 + TILE_X, s.W), ty1 = min(ty0 + TILE_Y, s.H);
    int count = 0;
    // grid nodes (iy, ix) in [iy0, iy1] x [ix0, ix1] (row major = ascending k, the order of the full scan)
    int iy0 = 0, ix0 = 0, nry = 0, nrx = 1, total = s.K;
    if (local) {
        auto floor_div = [](int a, int b) { return a >= 0 ? a / b : -((-a + b - 1) / b); };
        const int ny = s.K / s.grid_nx;
        iy0 = max(0, floor_div(ty0 - reach_y - s.grid_y0 + s.grid_dy - 1, s.grid_dy));
        const int iy1 = min(ny - 1, floor_div(ty1 + reach_y - s.grid_y0, s.grid_dy));
        ix0 = max(0, floor_div(tx0 - reach_x - s.grid_x0 + s.grid_dx - 1, s.grid_dx));
        const int ix1 = min(s.grid_nx - 1, floor_div(tx1 + reach_x - s.grid_x0, s.grid_dx));
        nry = max(iy1 - iy0 + 1, 0);
        nrx = max(ix1 - ix0 + 1, 1);
        total = ix1 >= ix0 ? nry * nrx : 0;
    }
    for (int k0 = 0; k0 < total; k0 += 64) {
        int k = k0 + lane;
        bool hit = false;
        float key = 0.f;
        if (k < total) {
            int4 w;
            if (local) {
                const int jy = k / nrx;
                k = (iy0 + jy) * s.grid_nx + ix0 + (k - jy * nrx);
                w = s.win[k];
            } else {
                w = k < klds ? lds_win[k] : s.win[k];
            }
            hit = w.x < ty1 && w.y > ty0 && w.z < tx1 && w.w > tx0;
            // heuristic sort key: squared distance of the window centre to the tile centre
            float my = 0.5f * (float)(w.x + w.y) - 0.5f * (float)(ty0 + ty1);
            float mx = 0.5f * (float)(w.z + w.w) - 0.5f * (float)(tx0 + tx1);
            key = my * my + mx * mx;
        }
        unsigned long long m = __ballot(hit);
        if (hit) {
            int pos = count + __popcll(m & ((1ULL << lane) - 1ULL));
            if (pos < MAXC) {
                ck[wave][pos] = k;
                ckey[wave][pos] = key;
            }
        }
        count += __popcll(m);
    }
    // (single wave per tile from here on: LDS writes above are ordered with the reads below)
    __builtin_amdgcn_s_waitcnt(0);
    const bool overflow = count > max_cand;
    if (lane == 0) {
        tile_count[tile] = overflow ? -count : count;
        s.tile_info[tile].count = overflow ? -count : count;
    }
    if (overflow) return;
    // count <= 64: lane c owns candidate c from here on.  Sort key: lower bound of the distance over the
    // whole tile (ascending -- k_slic_assign_dot stops at the first candidate beyond its bound), ties by
    // the distance of the window centre to the tile centre.
    const bool have = lane < count;
    const int k = have ? ck[wave][lane] : 0;
    Cand cd;
    cd.cy = s.cy[k]; cd.cx = s.cx[k]; cd.cL = s.cL[k]; cd.ca = s.ca[k]; cd.cb = s.cb[k];
    cd.win = s.win[k];
    cd.k = k;
    const double sw = s.spatial_weight;
    // coordinates relative to the tile centre (row 16, column 32): |Y| <= 16, |X| <= 32 for every pixel
    const double ryc = cd.cy - (double)(ty0 + 16), rxc = cd.cx - (double)(tx0 + 32);
    double dy = fmax(fmax(-16.0 - ryc, ryc - 15.0), 0.0), dx = fmax(fmax(-32.0 - rxc, rxc - 31.0), 0.0);
    const float lbt = __double2float_rd((dy * dy + dx * dx) * sw * 0.999999);
    const float key2 = have ? ckey[wave][lane] : 0.f;
    if (have) clb[wave][lane] = lbt;
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();
    int rank = 0;
    for (int j = 0; j < count; ++j) {
        const float kj = clb[wave][j], k2j = ckey[wave][j];
        rank += (kj < lbt) || (kj == lbt && (k2j < key2 || (k2j == key2 && j < lane)));
    }
    // reference colour of the tile: the colour of the first candidate (exact)
    const unsigned long long first = __ballot(have && rank == 0);
    const int src = first ? __ffsll((long long)first) - 1 : 0;
    const double ref0 = __shfl(cd.cL, src, 64), ref1 = __shfl(cd.ca, src, 64), ref2 = __shfl(cd.cb, src, 64);
    const double c0 = cd.cL - ref0, c1 = cd.ca - ref1, c2 = cd.cb - ref2;
    Rec32 rc;
    uint32_t rows_mask;
    rc.q0 = (float)(sw * (ryc * ryc + rxc * rxc) + (c0 * c0 + c1 * c1 + c2 * c2));
    rc.qy = (float)(-2.0 * sw * ryc);
    rc.qx = (float)(-2.0 * sw * rxc);
    rc.qL = (float)(-2.0 * c0);
    rc.qa = (float)(-2.0 * c1);
    rc.qb = (float)(-2.0 * c2);
    rc.lbt = lbt;
    {
        const int rlo = min(max(cd.win.x - ty0, 0), TILE_Y), rhi = min(max(cd.win.y - ty0, 0), TILE_Y);
        const int xlo = min(max(cd.win.z - tx0, 0), TILE_X), xhi = min(max(cd.win.w - tx0, 0), TILE_X);
        // bit 23 (xlo <= 64 leaves it free): the window spans the tile's columns inside the image -- with the row mask below the
        // whole window test of k_slic_assign_dot's common case is two scalar compares
        const uint32_t spans_x = (xlo == 0 && xhi >= tx1 - tx0) ? (1u << 23) : 0u;
        rc.meta = (uint32_t)rlo | ((uint32_t)rhi << 8) | ((uint32_t)xlo << 16) | ((uint32_t)xhi << 24) | spans_x;
        rows_mask = (rhi >= 32 ? 0xffffffffu : (1u << rhi) - 1u) & ~((1u << rlo) - 1u);          // (rlo <= rhi <= 32)
    }
    // fp32 copies of the older record layout (exact / first-sweep paths)
    cd.ry = (float)(cd.cy - (double)ty0);
    cd.rx = (float)(cd.cx - (double)tx0);
    cd.fL = (float)cd.cL; cd.fa = (float)cd.ca; cd.fb = (float)cd.cb;
    cd.mdc = s.slico ? s.mdc[k] : 1.0;
    if (have) {
        tile_cands[(size_t)tile * MAXC + rank] = cd;
        s.tile_rec[(size_t)tile * MAXC + rank] = rc;
        s.tile_k[(size_t)tile * MAXC + rank] = k;
        s.tile_rows[(size_t)tile * MAXC + rank] = rows_mask;
    }
    float qm[5] = { have ? fabsf(rc.qy) : 0.f, have ? fabsf(rc.qx) : 0.f, have ? fabsf(rc.qL) : 0.f,
                    have ? fabsf(rc.qa) : 0.f, have ? fabsf(rc.qb) : 0.f };
#pragma unroll
    for (int j = 0; j < 5; ++j)
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) qm[j] = fmaxf(qm[j], __shfl_xor(qm[j], off, 64));
    if (lane == 0) {
        TileInfo &ti = s.tile_info[tile];
        ti.ref[0] = ref0; ti.ref[1] = ref1; ti.ref[2] = ref2;
        ti.Qy = qm[0]; ti.Qx = qm[1]; ti.QL = qm[2]; ti.Qa = qm[3]; ti.Qb = qm[4];
    }
}

// ---------------------------------------------------------------------------------------------
// assignment + fused accumulation
// ---------------------------------------------------------------------------------------------
// Geometry: a workgroup = 4 waves = a 64 x 16 pixel tile; a lane owns one pixel column and ROWS = 4
// rows, so a wave load of one row of one Lab plane is a single 512-byte coalesced request.  Two
// vertically adjacent workgroups share the candidate list of their 64 x 32 bin tile.
//
// Exact fp64 distance (operation order of _slic.pyx, 2-D, unit spacing -- this IS the contract):
//   dist_center = (dy*dy + dx*dx) * spatial_weight;  dist_color = ((dL*dL) + da*da) + db*db;
//   dist_center += dist_color;          ties -> lowest centroid index
//
// fp32 pre-selection.  Every candidate is first evaluated in fp32 (half the issue cost, branch
// free), tracking per pixel the best (b1, slot) and the second-best value b2.  With
//   u = 2^-24, M >= |any Lab value or centroid colour|, R = 2*step+1 (window half width),
//   E = 3R + 64 (fp32 error of tile-relative coordinate differences, in units of u),
//   G = sqrt(2*sw)*E + sqrt(3)*4*M + 16
// the fp32 value d32 and the real-arithmetic value D of the same formula obey
//   |d32 - D| <= u*G*(D + 1); the fp64 reference value differs from D by < 1e-14*(D + 1).
// With kappa = 2*u*(G+1) (factor 2 = safety):  b - a > kappa*(a + b + 2)  =>  candidate a beats b
// in fp64 as well.  Hence
//   * b2 clear of b1   -> the fp32 winner is the exact argmin                     (~99.99 % of pixels)
//   * b2 inside        -> that row is decided by the exact fp64 loop over all its candidates
// First sweep: all centroids sit on the integer grid with colour 0, which puts whole lines of pixels
// on exact ties; it has its own path (FIRST) that orders the candidates by the integer squared
// distance n = dy^2 + dx^2: d = fl(fl(n*sw) + |pixel|^2) is strictly increasing in n (sw >> ulp), so
// argmin d == argmin n with ties -> lowest index, exactly what the fp64 evaluation yields.
// Candidates skipped by the lower-bound test satisfy lb32 - b1 > kappa*(lb32 + b1 + 2) for every
// row, i.e. they can neither win nor lie inside the margin, so not tracking them is harmless.
constexpr int ROWS = 4;               // rows per lane
constexpr int WAVES = 4;              // waves per workgroup
constexpr int WG_Y = ROWS * WAVES;    // 16 rows per workgroup, two workgroups per bin tile
static_assert(2 * WG_Y == SLIC_TILE_Y, "tile geometry");

// pixels the assignment kernel could not accumulate through its LDS slots (uncovered pixels that keep
// their previous label, tiles without a candidate list): plain global atomics, rare
__global__ void __launch_bounds__(256)
k_slic_leftover(SlicState s, const double *__restrict__ lab, const int32_t *__restrict__ labels)
{
    zshift_state(s);
    ZSHIFT(lab, s.zs); ZSHIFT(labels, s.zs);
    const int count = *s.leftover_count;
    const size_t plane = (size_t)s.H * s.W;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < count; i += gridDim.x * blockDim.x) {
        int p = s.leftover[i];
        int k = labels[p];
        if (k < 0) continue;            // never assigned so far: counted nowhere (as in the oracle)
        int y = p / s.W, x = p - y * s.W;
        long long *a = s.acc + (size_t)k * 9;
        const double fscale = ldexp(1.0, fix_bits_of(*s.premax));
        atomic_add_i64(a + 0, 1);
        atomic_add_i64(a + 1, y);
        atomic_add_i64(a + 2, x);
        fix_add_global(a + 3, (long long)trunc(lab[p] * fscale));
        fix_add_global(a + 5, (long long)trunc(lab[plane + p] * fscale));
        fix_add_global(a + 7, (long long)trunc(lab[2 * plane + p] * fscale));
    }
}

// SLICO (_slic.pyx, slic_zero): after the centres have been recomputed, max_dist_color[k] grows to the largest
// colour distance of a pixel of segment k to its NEW centre.  A maximum is order independent, and non-negative
// doubles order like their bit patterns: integer atomicMax, exact.  The plain read in front keeps the atomics
// rare (the maxima settle after the first few thousand pixels).
__global__ void __launch_bounds__(256)
k_slico_update(SlicState s, const double *__restrict__ lab, const int32_t *__restrict__ labels)
{
    const size_t n = (size_t)s.H * s.W;
    for (size_t p = (size_t)blockIdx.x * 256 + threadIdx.x; p < n; p += (size_t)gridDim.x * 256) {
        const int k = labels[p];
        if (k < 0) continue;
        const double t0 = lab[p] - s.cL[k], t1 = lab[n + p] - s.ca[k], t2 = lab[2 * n + p] - s.cb[k];
        double col = t0 * t0;
        col = col + t1 * t1;
        col = col + t2 * t2;
        unsigned long long *m = reinterpret_cast<unsigned long long *>(s.mdc + k);
        if (__hip_atomic_load(m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned long long)__double_as_longlong(col))
            atomicMax(m, (unsigned long long)__double_as_longlong(col));
    }
}

__device__ __forceinline__ double exact_dist(const Cand &cd, double fy, double fx, double sw, double L, double A, double B,
                                             bool slico = false)
{
    const double ty = cd.cy - fy, tx = cd.cx - fx;
    double d = (ty * ty + tx * tx) * sw;
    const double t0 = L - cd.cL, t1 = A - cd.ca, t2 = B - cd.cb;
    double col = t0 * t0;
    col = col + t1 * t1;
    col = col + t2 * t2;
    if (slico) col = col / cd.mdc;       // _slic.pyx: dist_center += dist_color / max_dist_color[k]  (IEEE division)
    return d + col;
}

// exact argmin over the whole candidate list of one pixel row (wave-uniform y)
__device__ __forceinline__ int exact_row(const Cand *__restrict__ cand, int nc, int y, int x, double sw, double L,
                                         double A, double B, bool slico = false)
{
    double bd = DBL_MAX;
    int bs = -1, bk = 0x7fffffff;
    const double fy = (double)y, fx = (double)x;
    for (int c = 0; c < nc; ++c) {
        const int4 w = cand[c].win;
        if (y < w.x || y >= w.y) continue;
        const bool inx = (x >= w.z) && (x < w.w);
        const double d = exact_dist(cand[c], fy, fx, sw, L, A, B, slico);
        const int k = cand[c].k;
        if (inx && ((bd > d) || (bd == d && k < bk))) {
            bd = d;
            bs = c;
            bk = k;
        }
    }
    return bs;
}

// no candidate list for this tile (more than SLIC_MAXC centroids reach it): scan the whole table
// in ascending k, strict '>' keeps the lowest k.  Returns -(k + 2), or -1 if nothing covers the pixel.
__device__ __forceinline__ int exact_row_global(const SlicState &s, int y, int x, double L, double A, double B)
{
    double bd = DBL_MAX;
    int res = -1;
    const double fy = (double)y, fx = (double)x;
    for (int k = 0; k < s.K; ++k) {
        const int4 w = s.win[k];
        if (y < w.x || y >= w.y || x < w.z || x >= w.w) continue;
        Cand cd;
        cd.cy = s.cy[k]; cd.cx = s.cx[k]; cd.cL = s.cL[k]; cd.ca = s.ca[k]; cd.cb = s.cb[k];
        cd.mdc = s.slico ? s.mdc[k] : 1.0;
        const double d = exact_dist(cd, fy, fx, s.spatial_weight, L, A, B, s.slico != 0);
        if (bd > d) {
            bd = d;
            res = -(k + 2);
        }
    }
    return res;
}

// wave sum through DPP (row_shr 1/2/4/8, row_bcast 15/31): no LDS traffic; total valid in lane 63
__device__ __forceinline__ int wave_sum_dpp_lane63(int v)
{
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);
    return v;
}

// sum of an integer-valued double (|x| < 2^47) over the wave as int64: split into a 24-bit low chunk
// and a high chunk (both exact), reduce as int32 through DPP, recombine; valid in lane 63
__device__ __forceinline__ long long wave_sum_limb_lane63(double x)
{
    const double h = floor(x * (1.0 / 16777216.0));
    const double l = x - h * 16777216.0;
    const int hs = wave_sum_dpp_lane63((int)h);
    const int ls = wave_sum_dpp_lane63((int)l);
    return ((long long)hs << 24) + (long long)ls;
}

// Segmented reduction of the centroid sums, shared by both assignment kernels.  A lane owns one pixel column
// and ROWS rows; `best_s[r]` is the slot (index into the tile's candidate list) the pixel was assigned to and
// `pending` has one bit per row that takes part.  Each 16-lane row of the wave (a 16 x 4 pixel block) works on
// the TWO smallest slots still pending in it, so that one pass finishes 95 % of the blocks.  Per slot four
// values per lane -- the fixed-point colour sums trunc(v * 2^f) (common.h; integer-valued doubles, exact up to
// the 64 pixels of a block) and one packed geometry word (n | sum of row offsets << 8 | sum of column offsets
// << 17) -- i.e. eight values go through one transposed DPP reduction over the row, and its even lanes add
// the totals to the int64 LDS slots of the workgroup (columns 0 n, 1 sum y, 2 sum x, 3 / 5 / 7 colour sums).
__device__ __forceinline__ void accumulate_block_sums(int lane, int x, int wy0, const int (&best_s)[4], unsigned pending,
                                                      const double (&pL)[4], const double (&pA)[4], const double (&pB)[4],
                                                      double fscale, long long (*lacc)[9])
{
    constexpr int NONE = 0x7fffffff;
    const int col = lane & 15;
    // the fixed-point terms of the four pixels of this lane (exact integers as doubles) and their geometry words
    double tL[4], tA[4], tB[4];
    int gw[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        tL[r] = trunc(pL[r] * fscale);
        tA[r] = trunc(pA[r] * fscale);
        tB[r] = trunc(pB[r] * fscale);
        gw[r] = 1 + (r << 8) + (col << 17);
    }
    while (__any(pending != 0)) {
        int m = NONE;
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (pending & (1u << r)) m = min(m, best_s[r]);
        const int sa = row16_min_i32(m);                    // smallest pending slot of the block
        m = NONE;
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if ((pending & (1u << r)) && best_s[r] != sa) m = min(m, best_s[r]);
        const int sb = row16_min_i32(m);                    // second smallest (NONE: the block has one label)
        // branch free: a row contributes t * 1.0 or t * 0.0 through an fma (exact either way), so the compiler has
        // nothing to turn into copies of the eight accumulators
        double q[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
        int ga = 0, gb = 0;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const bool pend = (pending >> r) & 1u;
            const bool ia = pend && best_s[r] == sa, ib = pend && best_s[r] == sb;
            const double ma = ia ? 1.0 : 0.0, mb = ib ? 1.0 : 0.0;
            q[0] = fma(tL[r], ma, q[0]); q[1] = fma(tA[r], ma, q[1]); q[2] = fma(tB[r], ma, q[2]);
            q[4] = fma(tL[r], mb, q[4]); q[5] = fma(tA[r], mb, q[5]); q[6] = fma(tB[r], mb, q[6]);
            ga += ia ? gw[r] : 0;
            gb += ib ? gw[r] : 0;
            pending &= ~((unsigned)(ia || ib) << r);
        }
        q[3] = (double)ga;
        q[7] = (double)gb;
        const double tot = row16_reduce8_f64(q, lane);      // lane pair j holds the row total of q[j]
        const int j = ((lane >> 3) & 1) * 4 + ((lane >> 2) & 1) * 2 + ((lane >> 1) & 1);
        const int slot = j < 4 ? sa : sb;
        if ((lane & 1) == 0 && slot != NONE) {
            const long long tv = (long long)tot;
            if ((j & 3) < 3) {
                if (tv != 0) atomic_add_i64(&lacc[slot][3 + 2 * (j & 3)], tv);
            } else {
                const long long n = tv & 255, sy = (tv >> 8) & 511, sx = tv >> 17;
                atomic_add_i64(&lacc[slot][0], n);
                atomic_add_i64(&lacc[slot][1], n * wy0 + sy);
                atomic_add_i64(&lacc[slot][2], n * (x - col) + sx);
            }
        }
    }
}

// workgroup LDS slots -> global sums (colour sums split into the two global limbs)
__device__ __forceinline__ void flush_block_sums(const long long (*lacc)[9], int nc, const int *slot_k, long long *acc, int tid)
{
    for (int i = tid; i < nc * 9; i += 256) {
        const int c = i / 9, j = i - 9 * c;
        const long long v = lacc[c][j];
        if (v == 0 || (j >= 3 && ((j - 3) & 1))) continue;
        long long *dst = acc + (size_t)slot_k[c] * 9 + j;
        if (j < 3) atomic_add_i64(dst, v);
        else fix_add_global(dst, v);
    }
}

// ACCUM: also accumulate the centroid sums (all sweeps but the last)
template <bool ACCUM, bool FIRST>
__global__ void __launch_bounds__(256)
k_slic_assign(SlicState s, const double *__restrict__ lab, int32_t *__restrict__ labels,
              const Cand *__restrict__ tile_cands, const int *__restrict__ tile_count)
{
    zshift_state(s);
    ZSHIFT_NN(lab, s.zs); ZSHIFT_NN(labels, s.zs); ZSHIFT_NN(tile_cands, s.zs); ZSHIFT_NN(tile_count, s.zs);
    __shared__ long long lacc[MAXC][9];
    __shared__ int lk_old[MAXC];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);          // scalar: keeps the row tests on the SALU
    const int tile = (blockIdx.y >> 1) * gridDim.x + blockIdx.x;       // 64 x 32 bin tile
    const int tx0 = blockIdx.x * TILE_X;
    const int wy0 = blockIdx.y * WG_Y + wave * ROWS;                    // first row of this wave
    const int rel0 = (blockIdx.y & 1) * WG_Y + wave * ROWS;             // ... relative to the bin tile
    const size_t plane = (size_t)s.H * s.W;
    const Cand *__restrict__ cand = tile_cands + (size_t)tile * MAXC;

    const int x = tx0 + lane;
    const bool xin = x < s.W;
    const double sw = s.spatial_weight;

    // pixel loads first: their HBM latency overlaps the candidate-table set-up below
    double pL[ROWS], pA[ROWS], pB[ROWS];
    int best_s[ROWS];      // slot in the candidate list; -1: nothing covers the pixel; <= -2: centroid -(k+2)
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
        int y = wy0 + r;
        bool ok = xin && y < s.H;
        size_t p = (size_t)(ok ? y : 0) * s.W + (ok ? x : 0);
        pL[r] = lab[p];
        pA[r] = lab[plane + p];
        pB[r] = lab[2 * plane + p];
        best_s[r] = -1;
    }

    const int total = tile_count[tile];
    const bool overflow = total < 0;             // block-uniform
    const int nc = overflow ? 0 : total;
    // candidate table in registers: lane c holds the fp32 record of candidate c (nc <= 64); the hot
    // loop fetches fields with v_readlane -- no memory latency in the loop at all
    int4 my_win = make_int4(0, 0, 0, 0);
    float my_ry = 0.f, my_rx = 0.f, my_fL = 0.f, my_fa = 0.f, my_fb = 0.f;
    int my_iy = 0, my_ix = 0, my_k = 0;
    if (lane < nc) {
        my_win = cand[lane].win;
        if (FIRST) {
            my_iy = (int)cand[lane].cy;
            my_ix = (int)cand[lane].cx;
            my_k = cand[lane].k;
        }
        my_ry = cand[lane].ry; my_rx = cand[lane].rx;
        my_fL = cand[lane].fL; my_fa = cand[lane].fa; my_fb = cand[lane].fb;
    }
    if (ACCUM) {
        for (int i = tid; i < nc * 9; i += 256) (&lacc[0][0])[i] = 0;
        __syncthreads();
    }

    if (overflow) {
#pragma unroll
        for (int r = 0; r < ROWS; ++r) best_s[r] = exact_row_global(s, wy0 + r, x, pL[r], pA[r], pB[r]);
    } else if (!s.fast32) {
#pragma unroll
        for (int r = 0; r < ROWS; ++r) best_s[r] = exact_row(cand, nc, wy0 + r, x, sw, pL[r], pA[r], pB[r], s.slico != 0);
    } else if (FIRST) {
        // integer distances (see header): centroid positions are exact integers in the first sweep
        int bn[ROWS], bk[ROWS];
#pragma unroll
        for (int r = 0; r < ROWS; ++r) bn[r] = bk[r] = 0x7fffffff;
        for (int c = 0; c < nc; ++c) {
            const int4 w = make_int4(__builtin_amdgcn_readlane(my_win.x, c), __builtin_amdgcn_readlane(my_win.y, c),
                                     __builtin_amdgcn_readlane(my_win.z, c), __builtin_amdgcn_readlane(my_win.w, c));
            if (w.x >= wy0 + ROWS || w.y <= wy0) continue;
            const int ciy = __builtin_amdgcn_readlane(my_iy, c), cix = __builtin_amdgcn_readlane(my_ix, c);
            const int ck = __builtin_amdgcn_readlane(my_k, c);
            const bool inx = (x >= w.z) && (x < w.w);
            const int dx = cix - x;
            const int dx2 = dx * dx;
#pragma unroll
            for (int r = 0; r < ROWS; ++r) {
                const int y = wy0 + r;
                if (y < w.x || y >= w.y) continue;
                const int dy = ciy - y;
                const int n = dy * dy + dx2;
                if (inx && (n < bn[r] || (n == bn[r] && ck < bk[r]))) {
                    bn[r] = n;
                    bk[r] = ck;
                    best_s[r] = c;
                }
            }
        }
    } else if (s.debug & 1) {
        // (profiling aid) no candidate loop at all: pure load + store
#pragma unroll
        for (int r = 0; r < ROWS; ++r) best_s[r] = nc > 0 ? 0 : -1;
    } else {
        constexpr int PHASE1 = 4;            // candidates evaluated before the per-lane bound is formed
        const float INF = __builtin_inff();
        const float sw32 = (float)sw;
        const float kappa = s.kappa;
        float fL[ROWS], fA[ROWS], fB[ROWS], b1[ROWS], b2[ROWS];
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
            fL[r] = (float)pL[r];
            fA[r] = (float)pA[r];
            fB[r] = (float)pB[r];
            b1[r] = b2[r] = INF;
        }
        const float flane = (float)lane;
        const float frow0 = (float)rel0;
        float mb = INF;                      // upper bound of this lane's best distances (stale = still valid)
#define RL_I(v) __builtin_amdgcn_readlane((v), c)
#define RL_F(v) __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), c))
        for (int c = 0; c < nc; ++c) {
            const int4 w = make_int4(RL_I(my_win.x), RL_I(my_win.y), RL_I(my_win.z), RL_I(my_win.w));
            if (c == PHASE1) {
                mb = b1[0];
#pragma unroll
                for (int r = 1; r < ROWS; ++r) mb = fmaxf(mb, b1[r]);
            }
            if (w.x >= wy0 + ROWS || w.y <= wy0) continue;   // wave-uniform: no row of this wave in the window
            const bool inx = (x >= w.z) && (x < w.w);
            const float ry = RL_F(my_ry) - frow0;             // centroid row relative to this wave's first row
            const float tx = RL_F(my_rx) - flane;
            const float dx2 = tx * tx;
            if (c >= PHASE1) {
                // lower bound over this lane's rows: |ry - r| is smallest at the row nearest to ry
                float tyb = 0.f;
                if (ry < 0.f) tyb = ry;
                else if (ry > (float)(ROWS - 1)) tyb = ry - (float)(ROWS - 1);
                const float lb = (tyb * tyb + dx2) * sw32;
                if (!__any(inx && (lb - mb <= kappa * (lb + mb + 2.f)))) continue;
            }
            const float cL = RL_F(my_fL), ca = RL_F(my_fa), cb = RL_F(my_fb);
#define SLIC_EVAL32(r)                                                                    \
    {                                                                                     \
        const float ty = ry - (float)(r);                                                 \
        const float t0 = fL[r] - cL, t1 = fA[r] - ca, t2 = fB[r] - cb;                    \
        float d = (ty * ty + dx2) * sw32 + (t0 * t0 + t1 * t1 + t2 * t2);                 \
        d = inx ? d : INF;                                                                \
        const bool lt1 = d < b1[r];                                                       \
        b2[r] = lt1 ? b1[r] : fminf(b2[r], d);                                            \
        b1[r] = fminf(b1[r], d);                                                          \
        best_s[r] = lt1 ? c : best_s[r];                                                  \
    }
            if (w.x <= wy0 && w.y >= wy0 + ROWS) {
                // the window covers every row of this wave (the common case): one branch-free block,
                // the four independent rows interleave in the VALU pipeline
                SLIC_EVAL32(0) SLIC_EVAL32(1) SLIC_EVAL32(2) SLIC_EVAL32(3)
            } else {
#pragma unroll
                for (int r = 0; r < ROWS; ++r) {
                    const int y = wy0 + r;
                    if (y < w.x || y >= w.y) continue;        // wave-uniform
                    SLIC_EVAL32(r)
                }
            }
#undef SLIC_EVAL32
        }
#undef RL_I
#undef RL_F
        // near ties (second best inside the margin for some pixel of the row): exact fp64 loop
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
            if (s.debug & 2) break;          // (profiling aid)
            const bool near2 = best_s[r] >= 0 && b2[r] < INF && !(b2[r] - b1[r] > kappa * (b1[r] + b2[r] + 2.f));
            if (!__any(near2)) continue;
            int e = exact_row(cand, nc, wy0 + r, x, sw, pL[r], pA[r], pB[r]);
            if (near2) best_s[r] = e;
        }
    }

    // labels: a pixel no window covers keeps its previous assignment (nearest_segments persists in
    // _slic.pyx).  Such pixels, and the pixels of tiles without a candidate list, are queued for
    // k_slic_leftover, which adds them to the centroid sums with plain global atomics (rare).
    unsigned pending = 0;
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
        int y = wy0 + r;
        if (!(xin && y < s.H)) continue;
        int p = y * s.W + x;
        if (best_s[r] >= 0) {
            labels[p] = cand[best_s[r]].k;
            pending |= 1u << r;
            continue;
        }
        if (best_s[r] <= -2) labels[p] = -(best_s[r] + 2);
        if (ACCUM) s.leftover[atomicAdd(s.leftover_count, 1)] = p;
    }
    if (!ACCUM || (s.debug & 4)) return;

    accumulate_block_sums(lane, x, wy0, best_s, pending, pL, pA, pB, ldexp(1.0, fix_bits_of(*s.premax)), lacc);
    __syncthreads();
    if (tid < nc) lk_old[tid] = cand[tid].k;
    __syncthreads();
    flush_block_sums(lacc, nc, lk_old, s.acc, tid);
}

// ---------------------------------------------------------------------------------------------
// assignment, second formulation (all sweeps but the first, fast path)
// ---------------------------------------------------------------------------------------------
// With Y, X the pixel position relative to the tile centre, f = p - ref its colour relative to the tile's
// reference colour and (ry, rx, c) the same for a centroid, the _slic.pyx distance is
//     D = sw*((ry-Y)^2 + (rx-X)^2) + |c - f|^2
//       = [sw*(ry^2+rx^2) + |c|^2] - 2*sw*ry*Y - 2*sw*rx*X - 2*c.f  +  [sw*(Y^2+X^2) + |f|^2]
//       =  q0 + qy*Y + qx*X + qL*fL + qa*fa + qb*fb                  +  P(pixel)
// P does not depend on the candidate, so the arg-min needs only the 5-term dot product d = D - P: one fma
// per lane and candidate for the X term and four fmas per pixel, issued as packed v_pk_fma_f32 on row
// pairs.  Coefficients come from k_slic_bin as 32-byte records read with scalar loads.
//
// Error bound (u = 2^-24).  Every coefficient and feature is the fp32 rounding of an exact fp64 value and
// the chain has five fmas, so |d32 - d*| <= 8*u*T with T = |q0| + |qy*Y| + |qx*X| + |qL*fL| + |qa*fa| +
// |qb*fb|.  With xb >= the sum of the five cross terms (|Y| <= 16, |X| <= 32, tile maxima of |q.|):
// q0 = d* - cross <= d* + xb, hence T <= d* + 2*xb, and for two candidates a, b of one pixel
//     |(d32_b - d32_a) - (D_b - D_a)| <= 8*u*(d_a + d_b + 4*xb).
// The kernel uses margin = 16*u*1.01*(b1 + b2 + 4*xb): a gap above it carries over to exact arithmetic
// with at least half the margin left, far above the ~1e-15 relative noise of the fp64 evaluation; anything
// closer sends the row to the exact fp64 loop (exact_row).
//
// Candidates are sorted by lbt = lower bound of D over the whole tile.  After PH1 (and again after PH2)
// candidates the wave forms W >= max over its pixels of (best D + margin); the first candidate with
// lbt > W ends the loop: it and all later ones can neither win nor come within the margin anywhere in the wave.
typedef float f2 __attribute__((ext_vector_type(2)));

// (profiling aid) cycles between successive marks of every wave, summed per phase
#define PHASE_MARK(j)                                                                              \
    if (PROF && s.phase_prof) {                                                                            \
        __builtin_amdgcn_s_waitcnt(0);                                                             \
        const long long now_ = (long long)__builtin_readcyclecounter();                            \
        if (tid == 0) prof_slot[j] += now_ - t_prev;                                               \
        t_prev = now_;                                                                             \
    }
#define PHASE_FLUSH()                                                                              \
    if (PROF && s.phase_prof && tid == 0) {                                                                \
        prof_slot[15] += 1;                                                                        \
        prof_slot[11] = (long long)wall_clock64();                                                 \
    }

// FIRST: first sweep.  All centroids sit on the regular grid (y0 + iy*dy, x0 + ix*dx) with colour 0, the
// distance is fl(fl(n*sw) + |p|^2) with n = dy^2 + dx^2 an integer, strictly increasing in n (sw >> ulp):
// the winner is the nearest grid node per axis, ties to the lower index (= lowest centroid index, what
// the strict '<' of _slic.pyx keeps).  The launcher only selects this variant when every pixel lies inside
// the search window of its nearest node.
#ifndef SLIC_DOT_MIN_BLOCKS
#define SLIC_DOT_MIN_BLOCKS 5
#endif
#ifndef SLIC_DOT_MIN_BLOCKS_TILE
#define SLIC_DOT_MIN_BLOCKS_TILE 4
#endif
// PROF: the instantiation with the in-kernel phase timers (IMSEGM_PHASE_PROF); the production kernels carry none of it
// (the marks end basic blocks and wait for all memory operations, which also keeps the compiler from overlapping phases)
template <bool ACCUM, bool FIRST, int U, bool PROF>
__global__ void __launch_bounds__(256, U == 1 ? SLIC_DOT_MIN_BLOCKS : SLIC_DOT_MIN_BLOCKS_TILE)
k_slic_assign_dot(SlicState s, const double *__restrict__ lab, int32_t *__restrict__ labels,
                  const Cand *__restrict__ tile_cands, const Rec32 *__restrict__ tile_rec,
                  const TileInfo *__restrict__ tile_info, const int *__restrict__ tile_k)
{
    zshift_state(s);
    ZSHIFT_NN(lab, s.zs); ZSHIFT_NN(labels, s.zs); ZSHIFT_NN(tile_cands, s.zs); ZSHIFT_NN(tile_rec, s.zs); ZSHIFT_NN(tile_info, s.zs); ZSHIFT_NN(tile_k, s.zs);
    __shared__ long long lacc[MAXC][9];
    __shared__ int lk[MAXC];
    long long t_prev = (PROF && s.phase_prof) ? (long long)__builtin_readcyclecounter() : 0;
    long long *prof_slot = s.phase_prof + ((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 2 + (ACCUM ? 1 : 0)) * 16;
    if (PROF && s.phase_prof && threadIdx.x == 0) {
        prof_slot[10] = (long long)wall_clock64();                     // start, 100 MHz (overwritten by every launch)
        prof_slot[12] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));    // HW_ID
        prof_slot[13] = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11));   // XCC_ID
    }

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tile_row = (blockIdx.y * U) >> 1;
    const int tile = tile_row * gridDim.x + blockIdx.x;                // 64 x 32 bin tile
    const int tx0 = blockIdx.x * TILE_X;
    // first row of this wave's unit 0 relative to the bin tile: U == 1: half a tile per workgroup, U == 2: the whole
    // tile, every wave works through two 64 x 4 units one after the other (pixel loads of both issued up front)
    const int relbase = U == 1 ? (blockIdx.y & 1) * WG_Y + wave * ROWS : wave * (ROWS * U);
    const size_t plane = (size_t)s.H * s.W;
    // (the tables come in as restrict-qualified kernel arguments so that their uniform reads are scalar loads)
    const Cand *__restrict__ cand = tile_cands + (size_t)tile * MAXC;
    const Rec32 *__restrict__ rec = tile_rec + (size_t)tile * MAXC;
    const TileInfo *__restrict__ ti = tile_info + tile;

    const int x = tx0 + lane;
    const bool xin = x < s.W;
    const double sw = s.spatial_weight;

#ifndef SLIC_SADDR
#define SLIC_SADDR 1
#endif
    double pLu[U][ROWS], pAu[U][ROWS], pBu[U][ROWS];
#if SLIC_SADDR
    // one scalar base per (row, plane) and ONE 32-bit byte offset per lane (global_load ... v_off, s[base]): the row index of a
    // unit is wave-uniform, so the 64-bit address arithmetic is scalar -- three vector instructions instead of fifty-five per wave.
    // A lane right of the image reads column 0 of its row, a row below the image reads row 0 (in bounds; never used).
    const unsigned xoff8 = (unsigned)(xin ? x : 0) * 8u;
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
            const int y = tile_row * TILE_Y + relbase + u * ROWS + r;
            const char *row = reinterpret_cast<const char *>(lab + (size_t)(y < s.H ? y : 0) * s.W);
            pLu[u][r] = *reinterpret_cast<const double *>(row + xoff8);
            pAu[u][r] = *reinterpret_cast<const double *>(row + plane * sizeof(double) + xoff8);
            pBu[u][r] = *reinterpret_cast<const double *>(row + 2 * plane * sizeof(double) + xoff8);
        }
#else
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
            const int y = tile_row * TILE_Y + relbase + u * ROWS + r;
            const bool ok = xin && y < s.H;
            const size_t p = (size_t)(ok ? y : 0) * s.W + (ok ? x : 0);
            pLu[u][r] = lab[p];
            pAu[u][r] = lab[plane + p];
            pBu[u][r] = lab[2 * plane + p];
        }
#endif
    const int my_k = tile_k[(size_t)tile * MAXC + lane];
    const unsigned my_rows = s.tile_rows[(size_t)tile * MAXC + lane];      // rows of the tile inside the slot's window (bit mask)
    // candidate table in registers: lane c holds the record of candidate c; the loop fetches the fields
    // with v_readlane, i.e. without any memory latency between two candidates
    const float4 my_ra = reinterpret_cast<const float4 *>(rec + lane)[0];      // q0, qx, qy, qL
    const float4 my_rb = reinterpret_cast<const float4 *>(rec + lane)[1];      // qa, qb, lbt, meta
    const int total = ti->count;
    const bool overflow = total < 0;             // block-uniform
    const int nc = overflow ? 0 : total;
    PHASE_MARK(0)                                  // all loads have landed
    if (ACCUM) {
        for (int i = tid; i < nc * 9; i += 256) (&lacc[0][0])[i] = 0;
        if (tid < MAXC) lk[tid] = my_k;
        __syncthreads();
    }
    PHASE_MARK(1)                                  // LDS clear + barrier

#pragma unroll
    for (int u = 0; u < U; ++u) {
    const int rel0 = relbase + u * ROWS;
    const int wy0 = tile_row * TILE_Y + rel0;                          // first row of this unit
    const double (&pL)[ROWS] = pLu[u], (&pA)[ROWS] = pAu[u], (&pB)[ROWS] = pBu[u];
    int best_s[ROWS];      // slot in the candidate list; -1: nothing covers the pixel; <= -2: centroid -(k+2)
#pragma unroll
    for (int r = 0; r < ROWS; ++r) best_s[r] = -1;

    if (overflow) {
#pragma unroll
        for (int r = 0; r < ROWS; ++r) best_s[r] = exact_row_global(s, wy0 + r, x, pL[r], pA[r], pB[r]);
    } else if (FIRST) {
        const int ny = s.K / s.grid_nx;
        auto nearest = [](int t, int d, int n) {             // node index nearest to offset t, ties -> lower
            int i = t <= 0 ? 0 : t / d;
            const int rem = t - i * d;
            if (t > 0 && 2 * rem > d) ++i;
            return min(i, n - 1);
        };
        const int ix = nearest(x - s.grid_x0, s.grid_dx, s.grid_nx);
        int want[ROWS];
#pragma unroll
        for (int r = 0; r < ROWS; ++r) want[r] = nearest(wy0 + r - s.grid_y0, s.grid_dy, ny) * s.grid_nx + ix;
        // centroid index -> slot of the tile list, one distinct index at a time
        unsigned todo = (1u << ROWS) - 1;
        while (true) {
            int mine = -1;
#pragma unroll
            for (int r = ROWS - 1; r >= 0; --r)
                if (todo & (1u << r)) mine = want[r];
            const unsigned long long vote = __ballot(mine >= 0);
            if (vote == 0) break;
            const int kt = __builtin_amdgcn_readlane(mine, __ffsll((long long)vote) - 1);
            const unsigned long long hit = __ballot(my_k == kt && lane < nc);
            const int slot = hit ? __ffsll((long long)hit) - 1 : -(kt + 2);
#pragma unroll
            for (int r = 0; r < ROWS; ++r)
                if ((todo & (1u << r)) && want[r] == kt) {
                    best_s[r] = slot;
                    todo &= ~(1u << r);
                }
        }
    } else if (nc > 0) {
#ifndef SLIC_PH1
#define SLIC_PH1 4
#endif
#ifndef SLIC_PH2
#define SLIC_PH2 8
#endif
        // Round 5: the same arithmetic per candidate and the same margins, with the instruction count of everything AROUND the
        // five multiply-adds cut down -- the kernel is bound by instruction issue (823 vector + 443 scalar instructions per wave
        // before, counted in the ISA and by SQ_INSTS_*):
        //  * the running best / second best live in register PAIRS (f2): the break bound and the near-tie test are packed
        //    operations (the bound: 26 instead of 83 instructions per evaluation), the near-tie test of the four rows ends in ONE
        //    wave vote;
        //  * ONE path through the candidate body -- a candidate whose window does not cover the whole unit masks its distances with
        //    four selects behind a uniform branch instead of taking a second copy of the body (the two copies met in ten register
        //    moves per candidate);
        //  * which rows of the unit a window covers comes from a 32-bit row mask of the tile per candidate (k_slic_bin: tile_rows)
        //    and "spans the tile's columns" from bit 23 of the record's meta word: two scalar compares where the window was decoded
        //    and compared field by field (30 scalar instructions per candidate);
        //  * the first PH1 candidates, which every wave evaluates, come through the scalar cache (s_load_dwordx8) instead of seven
        //    v_readlane each.
        constexpr int PH1 = SLIC_PH1, PH2 = SLIC_PH2;
        const float INF = __builtin_inff();
        const float sw32 = (float)sw;
        const double ref0 = ti->ref[0], ref1 = ti->ref[1], ref2 = ti->ref[2];
        f2 fL[2], fA[2], fB[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            fL[h] = (f2){ (float)(pL[2 * h] - ref0), (float)(pL[2 * h + 1] - ref0) };
            fA[h] = (f2){ (float)(pA[2 * h] - ref1), (float)(pA[2 * h + 1] - ref1) };
            fB[h] = (f2){ (float)(pB[2 * h] - ref2), (float)(pB[2 * h + 1] - ref2) };
        }
        const float X = (float)(lane - TILE_X / 2);
        const float Y0 = (float)(rel0 - TILE_Y / 2);
        const f2 Yp[2] = { (f2){ Y0, Y0 + 1.f }, (f2){ Y0 + 2.f, Y0 + 3.f } };
        // bound of the cross terms of this pixel over all candidates of the tile (row pairs)
        f2 xbp[2];
        {
            const float base = 16.f * ti->Qy + 32.f * ti->Qx, QL = ti->QL, Qa = ti->Qa, Qb = ti->Qb;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                xbp[h].x = fmaf(QL, fabsf(fL[h].x), fmaf(Qa, fabsf(fA[h].x), fmaf(Qb, fabsf(fB[h].x), base)));
                xbp[h].y = fmaf(QL, fabsf(fL[h].y), fmaf(Qa, fabsf(fA[h].y), fmaf(Qb, fabsf(fB[h].y), base)));
            }
        }
        f2 b1p[2] = { (f2){ INF, INF }, (f2){ INF, INF } }, b2p[2] = { (f2){ INF, INF }, (f2){ INF, INF } };
        const int rows_valid = max(0, min(ROWS, s.H - wy0));    // (0: a wave of the last workgroup below the image)
        const unsigned rows_all = (1u << rows_valid) - 1u;      // the rows of this unit inside the image
        unsigned wbound = 0x7f800000u;           // float bits of the break threshold (uniform)
        int c_end = nc;
#define RL_F(v) __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), c))
#define F2S(v) ((f2){ (v), (v) })
#ifndef SLIC_REC_SGPR
#define SLIC_REC_SGPR 1
#endif
        // one candidate against the four rows of this unit; `rows4` = which of them lie inside its window (from k_slic_bin's row mask)
        auto evaluate = [&](const int c, const float q0, const float qx, const float qy, const float qL, const float qa, const float qb,
                            const unsigned meta, const unsigned rows4) __attribute__((always_inline)) {
            {
                const f2 e2 = F2S(fmaf(qx, X, q0));
                f2 d01 = __builtin_elementwise_fma(F2S(qy), Yp[0], e2);
                f2 d23 = __builtin_elementwise_fma(F2S(qy), Yp[1], e2);
                d01 = __builtin_elementwise_fma(F2S(qL), fL[0], d01);
                d23 = __builtin_elementwise_fma(F2S(qL), fL[1], d23);
                d01 = __builtin_elementwise_fma(F2S(qa), fA[0], d01);
                d23 = __builtin_elementwise_fma(F2S(qa), fA[1], d23);
                d01 = __builtin_elementwise_fma(F2S(qb), fB[0], d01);
                d23 = __builtin_elementwise_fma(F2S(qb), fB[1], d23);
                if (!(rows4 == rows_all && (meta & (1u << 23)))) {
                    // the window does not cover every pixel of this unit: the pixels outside it do not see this candidate
                    const int xlo = (meta >> 16) & 0x7f, xhi = meta >> 24;
                    const bool inx = lane >= xlo && lane < xhi;
                    d01.x = (inx && (rows4 & 1u)) ? d01.x : INF;
                    d01.y = (inx && (rows4 & 2u)) ? d01.y : INF;
                    d23.x = (inx && (rows4 & 4u)) ? d23.x : INF;
                    d23.y = (inx && (rows4 & 8u)) ? d23.y : INF;
                }
#define SLIC_SELECT(r, b1v, b2v, dval)                                                             \
    {                                                                                              \
        const float d_ = (dval);                                                                   \
        const bool lt_ = d_ < (b1v);                                                               \
        (b2v) = __builtin_amdgcn_fmed3f((b1v), (b2v), d_);                                         \
        (b1v) = lt_ ? d_ : (b1v);                                                                  \
        best_s[r] = lt_ ? c : best_s[r];                                                           \
    }
                SLIC_SELECT(0, b1p[0].x, b2p[0].x, d01.x) SLIC_SELECT(1, b1p[0].y, b2p[0].y, d01.y)
                SLIC_SELECT(2, b1p[1].x, b2p[1].x, d23.x) SLIC_SELECT(3, b1p[1].y, b2p[1].y, d23.y)
#undef SLIC_SELECT
            }
        };
        int c_first = 0;
#if SLIC_REC_SGPR
        // the first PH1 candidates are evaluated by every wave (no bound exists before them): their records come through the
        // scalar cache (uniform address: s_load_dwordx8) instead of seven v_readlane each -- the vector unit is the bottleneck
        {
            const unsigned *__restrict__ rows_tab = s.tile_rows + (size_t)tile * MAXC;
#pragma unroll
            for (int c = 0; c < PH1; ++c) {
                if (c >= nc) break;
                const unsigned rows4 = (rows_tab[c] >> rel0) & 15u;
                if (rows4 == 0) continue;
                const Rec32 r = rec[c];
                evaluate(c, r.q0, r.qx, r.qy, r.qL, r.qa, r.qb, r.meta, rows4);
            }
            c_first = min(nc, PH1);
        }
#endif
        for (int c = c_first; c < nc; ++c) {
            if (c == PH1 || c == PH2) {
                // W >= max over the pixels of this wave of (best D + margin), D = d + P with the pixel's own term
                // P = sw * (Y^2 + X^2) + |f|^2; the factors 1.002 / 0.002 * (xb + 1) cover the fp32 evaluation error and the
                // near-tie margin many times over, so neither the order of the additions nor a pixel outside the image that
                // is counted in (a looser bound: the loop ends later, never earlier than it may) matters for the result
                float wl = 0.f;
                if (rows_valid == ROWS) {
                    // (everything below is loop invariant but for b1: the two opaque copies keep the compiler from hoisting it out
                    // of the loop into eight registers of every wave -- the registers decide between five and six waves per SIMD)
                    float Xv = X, c002 = 0.002f;
                    asm volatile("" : "+v"(Xv), "+v"(c002));
                    const float X2 = Xv * Xv;
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const f2 sp = __builtin_elementwise_fma(Yp[h], Yp[h], F2S(X2));
                        f2 P = F2S(sw32) * sp;
                        P = __builtin_elementwise_fma(fB[h], fB[h], P);
                        P = __builtin_elementwise_fma(fA[h], fA[h], P);
                        P = __builtin_elementwise_fma(fL[h], fL[h], P);
                        f2 t = b1p[h] + P;
                        t.x = fmaxf(t.x, 0.f);
                        t.y = fmaxf(t.y, 0.f);
                        const f2 v = __builtin_elementwise_fma(t, F2S(1.002f), __builtin_elementwise_fma(F2S(c002), xbp[h], F2S(c002)));
                        wl = fmaxf(fmaxf(v.x, v.y), wl);
                    }
                    wl = xin ? wl : 0.f;
                } else {
                    // the unit hangs over the lower edge of the image (H not a multiple of 4: one row of waves): no early end --
                    // a second form of the bound here is hoisted out of the loop by the compiler into every wave's preamble
                    wl = INF;
                }
                int wi = __float_as_int(wl);                  // wl >= 0: integer order == float order
                wi = max(wi, __builtin_amdgcn_update_dpp(0, wi, 0x111, 0xf, 0xf, false));
                wi = max(wi, __builtin_amdgcn_update_dpp(0, wi, 0x112, 0xf, 0xf, false));
                wi = max(wi, __builtin_amdgcn_update_dpp(0, wi, 0x114, 0xf, 0xf, false));
                wi = max(wi, __builtin_amdgcn_update_dpp(0, wi, 0x118, 0xf, 0xf, false));
                wi = max(wi, __builtin_amdgcn_update_dpp(0, wi, 0x142, 0xa, 0xf, false));
                wi = max(wi, __builtin_amdgcn_update_dpp(0, wi, 0x143, 0xc, 0xf, false));
                wbound = (unsigned)__builtin_amdgcn_readlane(wi, 63);
            }
            if ((unsigned)__builtin_amdgcn_readlane(__float_as_int(my_rb.z), c) > wbound) {
                c_end = c;
                break;
            }
            const unsigned rows4 = ((unsigned)__builtin_amdgcn_readlane((int)my_rows, c) >> rel0) & 15u;
            if (rows4 == 0) continue;
            evaluate(c, RL_F(my_ra.x), RL_F(my_ra.y), RL_F(my_ra.z), RL_F(my_ra.w), RL_F(my_rb.x), RL_F(my_rb.y),
                     (unsigned)__builtin_amdgcn_readlane(__float_as_int(my_rb.w), c), rows4);
        }
#undef RL_F
        PHASE_MARK(2)                              // candidate loop
        if (PROF && s.phase_prof && tid == 0) prof_slot[9] += c_end;
        // near ties (second best inside the margin for some pixel of the row): exact fp64 evaluation, in the
        // order of _slic.pyx, of the candidates whose fp32 value lies within the margin of the fp32 best --
        // the exact winner is always one of them (its d32 exceeds b1 by at most half the margin).
        // (b2 finite implies b1 finite implies a slot in best_s; b2 infinite: gap and margin are both infinite, hence the test)
        const float U16 = 16.f * 5.9604644775390625e-8f * 1.01f;
        f2 mg[2];
        bool near_r[ROWS];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            mg[h] = __builtin_elementwise_fma(F2S(U16), __builtin_elementwise_fma(F2S(4.f), xbp[h], b1p[h] + b2p[h]), F2S(1e-30f));
            const f2 gap = b2p[h] - b1p[h];
            near_r[2 * h] = b2p[h].x < INF && !(gap.x > mg[h].x);
            near_r[2 * h + 1] = b2p[h].y < INF && !(gap.y > mg[h].y);
        }
        if (__any(near_r[0] || near_r[1] || near_r[2] || near_r[3])) {
#pragma unroll
            for (int r = 0; r < ROWS; ++r) {
                const bool near2 = near_r[r];
                if (!__any(near2)) continue;
                const float m = r & 1 ? mg[r >> 1].y : mg[r >> 1].x, b1r = r & 1 ? b1p[r >> 1].y : b1p[r >> 1].x;
                const float l = r & 1 ? fL[r >> 1].y : fL[r >> 1].x, a = r & 1 ? fA[r >> 1].y : fA[r >> 1].x,
                            b = r & 1 ? fB[r >> 1].y : fB[r >> 1].x;
                const float Yr = Y0 + (float)r;
                const double fy = (double)(wy0 + r), fx = (double)x;
                double bd = DBL_MAX;
                int bs = -1, bk = 0x7fffffff;
#define RL_F(v) __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), c))
                for (int c = 0; c < c_end; ++c) {
                    const unsigned meta = (unsigned)__builtin_amdgcn_readlane(__float_as_int(my_rb.w), c);
                    const int rlo = meta & 0xff, rhi = (meta >> 8) & 0xff, xlo = (meta >> 16) & 0x7f, xhi = meta >> 24;
                    if (rel0 + r < rlo || rel0 + r >= rhi) continue;
                    float d = fmaf(RL_F(my_ra.y), X, RL_F(my_ra.x));
                    d = fmaf(RL_F(my_ra.z), Yr, d);
                    d = fmaf(RL_F(my_ra.w), l, d);
                    d = fmaf(RL_F(my_rb.x), a, d);
                    d = fmaf(RL_F(my_rb.y), b, d);
                    const bool take = near2 && lane >= xlo && lane < xhi && d - b1r <= m;
                    if (!__any(take)) continue;
                    const double e = exact_dist(cand[c], fy, fx, sw, pL[r], pA[r], pB[r]);
                    const int k = cand[c].k;
                    if (take && ((bd > e) || (bd == e && k < bk))) {
                        bd = e;
                        bs = c;
                        bk = k;
                    }
                }
#undef RL_F
                if (near2) best_s[r] = bs;
            }
        }
#undef F2S
        PHASE_MARK(3)                              // near-tie resolution
    }

    // labels: a pixel no window covers keeps its previous assignment (nearest_segments persists in
    // _slic.pyx).  Such pixels, and the pixels of tiles without a candidate list, are queued for
    // k_slic_leftover, which adds them to the centroid sums with plain global atomics (rare).
    unsigned pending = 0;
    int win_k[ROWS];
#pragma unroll
    for (int r = 0; r < ROWS; ++r) win_k[r] = __shfl(my_k, best_s[r] & 63, 64);
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
        const int y = wy0 + r;
        if (!(xin && y < s.H)) continue;
#if SLIC_SADDR
        int32_t &label_here = *reinterpret_cast<int32_t *>(reinterpret_cast<char *>(labels + (size_t)y * s.W) + (unsigned)x * 4u);
#else
        int32_t &label_here = labels[y * s.W + x];
#endif
        if (best_s[r] >= 0) {
            label_here = win_k[r];
            pending |= 1u << r;
            continue;
        }
        if (best_s[r] <= -2) label_here = -(best_s[r] + 2);
        if (ACCUM) {
            // no slot in the tile's list (tile without a list, or a pixel no window covers, which keeps its
            // previous label): straight to the global sums -- rare
            const int k = best_s[r] <= -2 ? -(best_s[r] + 2) : label_here;
            if (k >= 0) {
                // (with the centroid update inside this kernel such a contribution is not covered by the arrival count of its
                // centroid: the host is told and redoes the sweeps with separate finalize launches)
                if (s.fuse_finalize) *reinterpret_cast<volatile int *>(s.fail_host) = 1;
                long long *a = s.acc + (size_t)k * 9;
                const double fs = ldexp(1.0, fix_bits_of(*s.premax));
                atomic_add_i64(a + 0, 1);
                atomic_add_i64(a + 1, y);
                atomic_add_i64(a + 2, x);
                fix_add_global(a + 3, (long long)trunc(pL[r] * fs));
                fix_add_global(a + 5, (long long)trunc(pA[r] * fs));
                fix_add_global(a + 7, (long long)trunc(pB[r] * fs));
            }
        }
    }
    PHASE_MARK(4)                                  // label stores
    if (ACCUM) {
        accumulate_block_sums(lane, x, wy0, best_s, pending, pL, pA, pB, ldexp(1.0, fix_bits_of(*s.premax)), lacc);
        PHASE_MARK(5)                              // accumulation passes
    }
    }   // units
    if (!ACCUM) {
        PHASE_FLUSH()
        return;
    }
    __syncthreads();
    PHASE_MARK(6)                                  // barrier
    flush_block_sums(lacc, nc, lk, s.acc, tid);
    PHASE_MARK(7)                                  // flush
    PHASE_FLUSH()
    if (U == 1 && s.fuse_finalize) {
        // arrivals: every workgroup whose tile the search window of centroid k meets has k in its list (two workgroups per
        // 64 x 32 tile) and counts itself once its sums are through; the last one has all pixels of k in the global sums and
        // does what k_centroid_finalize does -- the table is read by the NEXT launch (k_slic_bin), so plain stores will do
        // Ordering, stated for gfx9 hardware rather than derived from the C++ memory model: the sums above are
        // device-scope atomics, performed at the L2 / memory side; `vmcnt` only reaches zero when every one of them has been
        // acknowledged from there, the barrier extends that to the whole workgroup, and the arrival below is again a device-scope
        // atomic -- so the workgroup that sees the last arrival reads complete sums with its device-scope loads.  An
        // acquire-release arrival would add a write-back of this CU's dirty label lines (16 MB per launch) to every workgroup's
        // tail for nothing: the labels are not part of the hand-over.  gfx950 is the only target of this library.
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (wave == 0 && lane < nc) {
            const int k = my_k;
            const int4 wv = cand[lane].win;
            const int expect = 2 * ((wv.y - 1) / TILE_Y - wv.x / TILE_Y + 1) * ((wv.w - 1) / TILE_X - wv.z / TILE_X + 1);
            const int seen = __hip_atomic_fetch_add(s.done + k, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1;
            if (seen == expect) {
                unsigned long long *a = reinterpret_cast<unsigned long long *>(s.acc + (size_t)k * 9);
                long long v[9];
#pragma unroll
                for (int j = 0; j < 9; ++j) v[j] = (long long)__hip_atomic_load(a + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (v[0] == 0) {
                    s.win[k] = make_int4(0, 0, 0, 0);          // no pixel carries label k any more: dead from now on
                } else {
                    const double nn = (double)v[0];
                    const double cy = i64_to_double(v[1]) / nn, cx = i64_to_double(v[2]) / nn;
                    const double finv = ldexp(1.0, -fix_bits_of(*s.premax));
                    s.cy[k] = cy;
                    s.cx[k] = cx;
                    s.cL[k] = fix_value(v[3], v[4], finv) / nn;
                    s.ca[k] = fix_value(v[5], v[6], finv) / nn;
                    s.cb[k] = fix_value(v[7], v[8], finv) / nn;
                    s.win[k] = search_window(cy, cx, s.step_y, s.step_x, s.H, s.W);
                    const int iy = k / s.grid_nx, ix = k - iy * s.grid_nx;
                    const double dy = fabs(cy - (double)(s.grid_y0 + iy * s.grid_dy)), dx = fabs(cx - (double)(s.grid_x0 + ix * s.grid_dx));
                    const int d = (int)ceil(fmax(dy, dx));
                    if (d > 0) atomicMax(&s.drift[s.drift_slot_next], d);
                }
#pragma unroll
                for (int j = 0; j < 9; ++j) s.acc[(size_t)k * 9 + j] = 0;
                s.done[k] = 0;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// the launches of the sweeps
// ---------------------------------------------------------------------------------------------
// (Round 3 built every sweep after the first as ONE persistent launch -- k_slic_sweeps: work items (sweep, tile) pulled from queues
// by resident workgroups, per-row completion counters, records handed from sweep to sweep through agent-scope loads -- bit-exact
// and not faster: 514 us against ~480 us + launch boundaries for one image, 1.23 against 0.76 ms per image with three in flight
// (profiles/bench_r03_persistent_*.json, profiles/HISTORY.md "Round 3").  It stayed opt-in for three rounds and went in round 6.)
static std::atomic<long> g_sweep_fallback{0};
void slic_sweep_counters(long *persistent_runs, long *fallback_runs)
{
    if (persistent_runs) *persistent_runs = 0;         // (the one-launch sweeps of round 3 are gone)
    if (fallback_runs) *fallback_runs = g_sweep_fallback.load();
}
void slic_sweep_note_fallback() { g_sweep_fallback.fetch_add(1); }

// function attributes of the sweep kernels on the current device (idempotent; must not run inside a stream capture)
int slic_prepare_device()
{
    static bool bin_attr[IMSEGM_MAX_DEVICES];           // per device, like every function attribute
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    if (dev < 0 || dev >= IMSEGM_MAX_DEVICES || !bin_attr[dev]) {
        HIP_TRY(hipFuncSetAttribute((const void *)k_slic_bin, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    BIN_MAX_K_LDS * (int)sizeof(int4)));
        if (dev >= 0 && dev < IMSEGM_MAX_DEVICES) bin_attr[dev] = true;
    }
    return 0;
}

int launch_slic_iterations(SlicState s, const double *lab, const double *init_yx_dev, int32_t *labels, int max_iter,
                           int max_cand, const ProfHook &prof, hipStream_t st, bool *fused_update, ZBatch zb)
{
    if (fused_update) *fused_update = false;
    // several images per launch: image b = blockIdx.z, every device pointer of `s` (and lab / labels) b * zb.zs bytes further on;
    // the failure word of the fused centroid update is ONE host word for the whole batch
    s.zs = zb.zs;
    const unsigned nz = (unsigned)zb.nz;
    if (nz > 1 && (s.slico || s.phase_prof)) {
        set_error("slic: SLICO and the phase profiler do not take a batch");
        return -1;
    }
    const bool default_cand = max_cand <= 0 || max_cand >= MAXC;
    // Centroid update inside the assignment kernel (the workgroup that completes a centroid divides its sums): measured on
    // MI355X it wins where the sweep is launch bound -- the 647 x 1024 images of config 4: +8 % images/s -- and costs the
    // assignment kernel 3 us of tail at 2048 x 2048 (42.2 against 39.1 us, the stage as a whole 6 us faster), so it is the default
    // only when the whole assignment grid is resident at once (one generation of workgroups: nothing hides the extra launch).
    const bool env_separate_finalize = knobs().separate_finalize;      // (tests switch these: imsegm_debug_reload_env)
    const bool env_fuse_finalize = knobs().fuse_finalize;
    const long assign_workgroups = 2L * cdiv(s.W, TILE_X) * cdiv(s.H, TILE_Y);
    // (a list capacity below the default leaves tiles without a list, whose pixels bypass the arrival count: every image would
    // be handed back and run twice -- such runs keep the separate finalize launches)
    const bool fail_host_fuse = s.done != nullptr && s.fail_host != nullptr && !env_separate_finalize && default_cand &&
                                (env_fuse_finalize || assign_workgroups <= 1280);
    if (fail_host_fuse) *s.fail_host = 0;
    if (max_cand <= 0 || max_cand > MAXC) max_cand = MAXC;
    size_t n = (size_t)s.H * s.W;
    hipLaunchKernelGGL(k_centroid_init, dim3(cdiv(s.K, 256), 1, nz), 256, 0, st, s, init_yx_dev);
    dim3 grid(cdiv(s.W, TILE_X), 2 * cdiv(s.H, TILE_Y), nz);     // two 64 x 16 workgroups per bin tile
    const int n_tiles = grid.x * cdiv(s.H, TILE_Y);
    // workgroups of the dot kernel: 1 = half a bin tile (64 x 16), 2 = a whole tile, two units per wave
    const dim3 grid_tile(grid.x, cdiv(s.H, TILE_Y), nz);
    const int units = s.assign_units == 2 ? 2 : 1;
    if (slic_prepare_device()) return -1;
    // first sweep in closed form: allowed when every pixel lies inside the search window of its nearest grid
    // node (per axis: half a grid step in the interior, the border offsets at the ends)
    bool grid_covers = false;
    {
        const int ny = s.K / s.grid_nx;
        const int reach_y = std::max(std::max(s.grid_y0, s.H - 1 - (s.grid_y0 + (ny - 1) * s.grid_dy)), s.grid_dy / 2 + 1);
        const int reach_x = std::max(std::max(s.grid_x0, s.W - 1 - (s.grid_x0 + (s.grid_nx - 1) * s.grid_dx)), s.grid_dx / 2 + 1);
        grid_covers = reach_y <= 2 * s.step_y && reach_x <= 2 * s.step_x;
    }
    // nearest = -1, unless the first sweep is the closed-form one, which gives every pixel a label without looking at the map
    if (!(max_iter > 0 && s.fast32 && s.spatial_weight > 1e-9 && grid_covers && !(s.debug & 32)))
        for (unsigned b = 0; b < nz; ++b)
            HIP_TRY(hipMemsetAsync(reinterpret_cast<unsigned char *>(labels) + (size_t)b * zb.zs, 0xff, n * sizeof(int32_t), st));
    static const bool print_occ = getenv("IMSEGM_PRINT_OCC") != nullptr;
    if (print_occ) {
        int nb = 0;
        HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void *)k_slic_assign_dot<true, false, 1, false>, 256, 0));
        fprintf(stderr, "[occupancy] k_slic_assign_dot<true>: %d workgroups per CU\n", nb);
        HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void *)k_slic_assign_dot<false, false, 1, false>, 256, 0));
        fprintf(stderr, "[occupancy] k_slic_assign_dot<false>: %d workgroups per CU\n", nb);
    }
    for (int it = 0; it < max_iter; ++it) {
        hipLaunchKernelGGL(k_slic_bin, dim3(cdiv(n_tiles, BIN_TILES_PER_BLOCK), 1, nz), 256,
                           (size_t)std::min(s.K, BIN_MAX_K_LDS) * sizeof(int4), st, s, (int)grid.x, n_tiles, max_cand,
                           s.tile_cands, s.tile_count, it % SLIC_DRIFT_SLOTS);
        // first sweep: integer-grid centroids with zero colour -> exact integer path (needs the
        // fast-path preconditions and a spatial weight far above the fp64 resolution)
        const bool first = it == 0 && s.fast32 && s.spatial_weight > 1e-9;
        // SLICO: the first sweep is the plain one (colour 0, maxima 1: the colour term is common to all
        // candidates); every later sweep divides by a per-centroid maximum and takes the exact fp64 path
        const bool dot = !first && s.fast32 && !s.slico && !(s.debug & 16);
        const bool first_grid = first && grid_covers && !(s.debug & 32);
        const bool accum = it + 1 < max_iter;
        // centroid update inside the assignment kernel of this sweep (no finalize launch behind it)
        const bool fuse = accum && s.done && fail_host_fuse && units == 1 && !s.phase_prof && (first_grid || dot);
        if (fuse && fused_update) *fused_update = true;             // (the caller reads the failure word after its next synchronisation)
        SlicState sf = s;
        sf.fuse_finalize = fuse ? 1 : 0;
        sf.drift_slot_next = (it + 1) % SLIC_DRIFT_SLOTS;
        // when profiling, the event pair rides on the dispatch itself (kernel begin / end timestamps)
        hipEvent_t ev_a = nullptr, ev_b = nullptr;
        if (prof.pair) prof.pair(prof.user, 0, &ev_a, &ev_b);
// (with a profiler pair: the extended launch that stamps the dispatch; otherwise a plain launch, which a stream capture records)
#define LAUNCH_ON(kernel, g, ...)                                                                                    \
    {                                                                                                                \
        if (ev_a) hipExtLaunchKernelGGL(kernel, g, dim3(256), 0, st, ev_a, ev_b, 0, __VA_ARGS__);                    \
        else hipLaunchKernelGGL(kernel, g, dim3(256), 0, st, __VA_ARGS__);                                           \
    }
#define LAUNCH_ASSIGN(kernel, ...) LAUNCH_ON(kernel, grid, __VA_ARGS__)
#define LAUNCH_DOT(ACC, FST)                                                                                         \
    {                                                                                                                \
        if (units == 2)                                                                                              \
            LAUNCH_ON((k_slic_assign_dot<ACC, FST, 2, false>), grid_tile, s, lab, labels, s.tile_cands, s.tile_rec, s.tile_info, s.tile_k) \
        else if (s.phase_prof)                                                                                       \
            LAUNCH_ON((k_slic_assign_dot<ACC, FST, 1, true>), grid, s, lab, labels, s.tile_cands, s.tile_rec, s.tile_info, s.tile_k) \
        else                                                                                                         \
            LAUNCH_ON((k_slic_assign_dot<ACC, FST, 1, false>), grid, sf, lab, labels, s.tile_cands, s.tile_rec, s.tile_info, s.tile_k) \
    }
        if (first_grid) {
            if (accum) LAUNCH_DOT(true, true) else LAUNCH_DOT(false, true)
        } else if (dot) {
            if (accum) LAUNCH_DOT(true, false) else LAUNCH_DOT(false, false)
        } else if (first) {
            if (accum) LAUNCH_ASSIGN((k_slic_assign<true, true>), s, lab, labels, s.tile_cands, s.tile_count)
            else LAUNCH_ASSIGN((k_slic_assign<false, true>), s, lab, labels, s.tile_cands, s.tile_count)
        } else {
            SlicState se = s;
            if (s.slico) se.fast32 = 0;
            if (accum) LAUNCH_ASSIGN((k_slic_assign<true, false>), se, lab, labels, s.tile_cands, s.tile_count)
            else LAUNCH_ASSIGN((k_slic_assign<false, false>), se, lab, labels, s.tile_cands, s.tile_count)
        }
#undef LAUNCH_DOT
#undef LAUNCH_ON
#undef LAUNCH_ASSIGN
        if (it + 1 < max_iter) {
            if (!dot && !first_grid) hipLaunchKernelGGL(k_slic_leftover, dim3(64, 1, nz), 256, 0, st, s, lab, labels);
            if (!fuse) hipLaunchKernelGGL(k_centroid_finalize, dim3(cdiv(s.K, 256), 1, nz), 256, 0, st, s, (it + 1) % SLIC_DRIFT_SLOTS);
            if (s.slico)
                hipLaunchKernelGGL(k_slico_update, (int)std::min<size_t>(cdiv(n, (size_t)256), 4096), 256, 0, st, s, lab, labels);
        }
    }
    HIP_TRY(hipGetLastError());
    return 0;
}

}  // namespace imsegm
