// texture.hip -- Leung-Malik texture responses on the GPU (BASELINE config 3).
//
// Replaces the scipy.ndimage calls of /root/reference/imsegm/descriptors.py:
//   :1078  img - gaussian_filter(img.astype(float), 150)   (scalar sigma on H, W AND the channel axis)
//   :951-978  compute_img_filter_response2d/3d: ndimage.convolve(img, fl) per kernel of a battery,
//             maximum over the orientations of a multi-kernel battery
//   :1088-1094  clip at MAX_SIGNAL_RESPONSE, global L2 norm over the three channels
// The per-superpixel statistics of the (rescaled) response then run in stats.hip.
//
// This is the one compute-bound piece of the path (2 * 1089 * 76 * 3 = 5e5 fp64 flop per pixel): the
// roofline is the fp64 vector FMA peak, not HBM.  Filter responses need only
// agree with the reference to 1e-5, so FMA contraction and any summation order are fine here.
#include "slic.h"

namespace imsegm {

__device__ __forceinline__ int reflect_index(int i, int n)
{
    // scipy.ndimage 'reflect' (half-sample symmetric), any distance
    if (n == 1) return 0;
    int p = 2 * n;
    i %= p;
    if (i < 0) i += p;
    if (i >= n) i = p - 1 - i;
    return i;
}

// image (interleaved, any supported dtype) -> three fp64 planes
template <typename T>
__global__ void __launch_bounds__(256) k_to_planes(const T *__restrict__ img, int n, double *__restrict__ planes)
{
    int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
#pragma unroll
    for (int c = 0; c < 3; ++c) planes[(size_t)c * n + p] = (double)img[3 * (size_t)p + c];
}

// ---- long symmetric 1-D correlation (sigma = 150 -> radius 600, 1201 taps) ------------------------------------------------
// Round 2 computed one output per lane with two global loads per FMA (3.5 TFLOP/s = 4.5 % of the fp64 vector peak, 17 % of a
// config-3 image's time).  Now a lane owns one column and CR = 16 consecutive output rows: every value it loads feeds 16 FMAs,
// and the 31 taps a block of 16 source rows needs are a contiguous window of the (zero padded) full tap table, wave uniform ->
// scalar loads, as the weights of the filter batteries.  The pass is always along y (coalesced row segments); the x pass runs
// on the transposed planes (two LDS-tiled transposes, 2 x 100 MB each, instead of strided reads).
constexpr int CR = 16;
// fullpad[CR - 1 + d + radius] = taps[|d|] for |d| <= radius, zero elsewhere (length 2 * radius + 1 + 2 * CR)
__global__ void k_taps_full(const double *__restrict__ taps, int radius, double *__restrict__ fullpad)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 2 * radius + 1 + 2 * CR) return;
    const int d = i - (CR - 1) - radius;
    fullpad[i] = (d >= -radius && d <= radius) ? taps[d < 0 ? -d : d] : 0.0;
}

__global__ void __launch_bounds__(256)
k_corr1d_col(const double *__restrict__ src, double *__restrict__ dst, int H, int W, const double *__restrict__ fullpad, int radius)
{
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int y0 = (blockIdx.y * 4 + wave) * CR;
    if (y0 >= H) return;
    const int x = blockIdx.x * 64 + lane;
    const int xs = x < W ? x : W - 1;
    const double *s = src + (size_t)blockIdx.z * H * W;
    double acc[CR];
#pragma unroll
    for (int i = 0; i < CR; ++i) acc[i] = 0.0;
    // source rows q = y0 - radius + b * CR + j; for output row y0 + i the tap is d = q - (y0 + i) = -radius + b * CR + j - i, i.e.
    // fullpad[(CR - 1) + radius + d] = fullpad[b * CR + (CR - 1) + j - i]: the window w[m], m = j - i + CR - 1, starts at b * CR
    const int nblocks = (2 * radius + CR + CR - 1) / CR;
    for (int b = 0; b < nblocks; ++b) {
        const double *wp = fullpad + (size_t)b * CR;
        double w[2 * CR - 1];
#pragma unroll
        for (int m = 0; m < 2 * CR - 1; ++m) w[m] = wp[m];
        double v[CR];
        const int q0 = y0 - radius + b * CR;            // (wave uniform)
        if (q0 >= 0 && q0 + CR <= H) {
            // all CR source rows inside the plane -- nearly every block: CR loads off one row pointer.  (The reflected row index
            // costs ~20 scalar instructions per row, and the scalar unit serves the four SIMDs of a CU: with it in every block the
            // pass ran at a quarter of the vector rate, 1.47 ms per pass at 2048^2.)
            const double *p = s + (size_t)q0 * W + xs;
#pragma unroll
            for (int j = 0; j < CR; ++j) v[j] = p[(size_t)j * W];
        } else {
#pragma unroll
            for (int j = 0; j < CR; ++j) v[j] = s[(size_t)reflect_index(q0 + j, H) * W + xs];
        }
#pragma unroll
        for (int j = 0; j < CR; ++j)
#pragma unroll
            for (int i = 0; i < CR; ++i) acc[i] = fma(w[j - i + CR - 1], v[j], acc[i]);
    }
    if (x < W) {
        double *d = dst + (size_t)blockIdx.z * H * W;
#pragma unroll
        for (int i = 0; i < CR; ++i)
            if (y0 + i < H) d[(size_t)(y0 + i) * W + x] = acc[i];
    }
}

// dst[p][x][y] = src[p][y][x]: 32 x 32 tiles through LDS (33 columns: no bank conflicts), P planes
__global__ void __launch_bounds__(256) k_transpose_planes(const double *__restrict__ src, double *__restrict__ dst, int H, int W)
{
    __shared__ double tile[32][33];
    const size_t plane = (size_t)blockIdx.z * H * W;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;          // 32 x 8 threads
    const int x0 = blockIdx.x * 32, y0 = blockIdx.y * 32;
#pragma unroll
    for (int r = 0; r < 32; r += 8) {
        const int x = x0 + tx, y = y0 + ty + r;
        if (x < W && y < H) tile[ty + r][tx] = src[plane + (size_t)y * W + x];
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 32; r += 8) {
        const int y = y0 + tx, x = x0 + ty + r;                      // transposed: the fast index runs along the old y
        if (x < W && y < H) dst[plane + (size_t)x * H + y] = tile[tx][ty + r];
    }
}

// blur along y then along x: src -> out, tmp: scratch of the same size (src is left untouched)
static int launch_long_blur(const double *src, double *tmp, double *out, int P, int H, int W, const double *fullpad, int radius,
                            hipStream_t st)
{
    hipLaunchKernelGGL(k_corr1d_col, dim3(cdiv(W, 64), cdiv(H, 4 * CR), P), 256, 0, st, src, tmp, H, W, fullpad, radius);      // y pass
    hipLaunchKernelGGL(k_transpose_planes, dim3(cdiv(W, 32), cdiv(H, 32), P), 256, 0, st, tmp, out, H, W);                    // -> W x H
    hipLaunchKernelGGL(k_corr1d_col, dim3(cdiv(H, 64), cdiv(W, 4 * CR), P), 256, 0, st, out, tmp, W, H, fullpad, radius);      // x pass
    hipLaunchKernelGGL(k_transpose_planes, dim3(cdiv(H, 32), cdiv(W, 32), P), 256, 0, st, tmp, out, W, H);                    // -> H x W
    HIP_TRY(hipGetLastError());
    return 0;
}

// gray volume: every slice is an independent plane (descriptors.py:981-994 image_subtract_gauss_smooth)
template <typename T>
__global__ void __launch_bounds__(256) k_vol_to_planes(const T *__restrict__ vol, size_t n, double *__restrict__ planes)
{
    size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p < n) planes[p] = (double)vol[p];
}
__global__ void __launch_bounds__(256)
k_subtract(const double *__restrict__ orig, const double *__restrict__ blur, size_t n, double *__restrict__ out)
{
    size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p < n) out[p] = orig[p] - blur[p];
}

// channel-axis pass of the 3-D Gaussian (a 3 x 3 mixing matrix, folded on the host) and the
// subtraction: out = original - blurred
__global__ void __launch_bounds__(256)
k_mix_subtract(const double *__restrict__ orig, const double *__restrict__ blur, int n, const double *__restrict__ mix,
               double *__restrict__ out)
{
    int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    double b[3] = { blur[p], blur[(size_t)n + p], blur[2 * (size_t)n + p] };
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        double m = mix[3 * c] * b[0] + mix[3 * c + 1] * b[1] + mix[3 * c + 2] * b[2];
        out[(size_t)c * n + p] = orig[(size_t)c * n + p] - m;
    }
}

// ---- filter battery ---------------------------------------------------------------------------------
// Workgroup = 64 x 16 output pixels of one channel plane; the (16 + 2r) x (64 + 2r) input tile sits in
// LDS.  A lane owns one output column and 4 rows; every LDS value it reads feeds up to 4 rows x NK
// kernels = 32 FMAs (NK = 8).  Weights are wave-uniform: stored as W[kx][t][k] so that the NK weights of
// one tap are one contiguous scalar load.
constexpr int CV_TX = 64, CV_TY = 16, CV_ROWS = 4;

// rows of the weight table per kernel column: S rows + the CV_ROWS - 1 zero rows the sliding window runs into, rounded up to
// whole groups of CV_ROWS (k_conv_battery unrolls its row loop in such groups)
__host__ __device__ static inline int conv_padded_rows(int radius) { return ((2 * radius + 1 + CV_ROWS - 1 + CV_ROWS - 1) / CV_ROWS) * CV_ROWS; }

// wgt: [kx][t][k] as the caller passes it (S x S x NK) -> [kx][t < Spad][k], zero rows behind t = S - 1
__global__ void k_pad_weights(const double *__restrict__ wgt, int S, int Spad, int nk, double *__restrict__ out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= S * Spad * nk) return;
    const int k = i % nk, t = (i / nk) % Spad, kx = i / (nk * Spad);
    out[i] = t < S ? wgt[((size_t)kx * S + t) * nk + k] : 0.0;
}

template <int NK>
__global__ void __launch_bounds__(256)
k_conv_battery(const double *__restrict__ planes, int H, int W, const double *__restrict__ wgt, int radius,
               double clip, double *__restrict__ resp)
{
    extern __shared__ double tile[];                 // [(CV_TY - CV_ROWS + Spad)][(CV_TX + 2r)]
    const int S = 2 * radius + 1, Spad = conv_padded_rows(radius);
    const int tw = CV_TX + 2 * radius, th = CV_TY + 2 * radius, th_pad = CV_TY - CV_ROWS + Spad;
    const int ch = blockIdx.z;
    const double *src = planes + (size_t)ch * H * W;
    const int x0 = blockIdx.x * CV_TX, y0 = blockIdx.y * CV_TY;
    for (int i = threadIdx.x; i < tw * th_pad; i += 256) {
        int ty = i / tw, tx = i - ty * tw;
        // (rows behind the halo only ever meet zero weights: any finite value will do)
        int gy = reflect_index(y0 + min(ty, th - 1) - radius, H), gx = reflect_index(x0 + tx - radius, W);
        tile[i] = src[(size_t)gy * W + gx];
    }
    __syncthreads();
    const int lx = threadIdx.x & 63;
    const int ly = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) * CV_ROWS;
    double acc[NK][CV_ROWS];
#pragma unroll
    for (int k = 0; k < NK; ++k)
#pragma unroll
        for (int i = 0; i < CV_ROWS; ++i) acc[k][i] = 0.0;
    // correlation form: out[y][x] = sum_{t, kx} Wc[t][kx] * in[y + t - r][x + kx - r]; the host passes the flipped kernels, so
    // this equals ndimage.convolve.  The input value of tile row ly + ky meets weight row t = ky - i for output row i: a window of
    // CV_ROWS weight rows slides down the kernel column, ONE new row of NK wave-uniform weights (a scalar load) per input value;
    // the row loop is unrolled in groups of CV_ROWS, so the window lives in fixed scalar registers without any moves.
    for (int kx = 0; kx < S; ++kx) {
        const double *wk = wgt + (size_t)kx * Spad * NK;
        double w[CV_ROWS][NK];
#pragma unroll
        for (int u = 0; u < CV_ROWS; ++u)
#pragma unroll
            for (int k = 0; k < NK; ++k) w[u][k] = 0.0;
        for (int g = 0; g < Spad; g += CV_ROWS) {
#pragma unroll
            for (int u = 0; u < CV_ROWS; ++u) {
                const int ky = g + u;
#pragma unroll
                for (int k = 0; k < NK; ++k) w[u][k] = wk[ky * NK + k];          // weight row t = ky into slot ky mod CV_ROWS
                const double v = tile[(ly + ky) * tw + lx + kx];
#pragma unroll
                for (int i = 0; i < CV_ROWS; ++i)
#pragma unroll
                    for (int k = 0; k < NK; ++k) acc[k][i] = fma(w[(u - i + CV_ROWS) % CV_ROWS][k], v, acc[k][i]);
            }
        }
    }
    const int x = x0 + lx;
#pragma unroll
    for (int i = 0; i < CV_ROWS; ++i) {
        const int y = y0 + ly + i;
        if (x >= W || y >= H) continue;
        double r = acc[0][i];
#pragma unroll
        for (int k = 1; k < NK; ++k) r = fmax(r, acc[k][i]);
        if (r > clip) r = clip;
        resp[(size_t)ch * H * W + (size_t)y * W + x] = r;
    }
}

// ---- dense batteries of kernels with a point symmetry ------------------------------------------------------------------------
// Every edge filter of the bank is odd and every bar filter even under the point reflection p -> -p, bit for bit (the rotated grid
// of descriptors.py:924-928 is negated exactly): K(-p) = sign * K(p), so
//     sum_p K(p) in(q + p)  =  sum_{p in half plane} K(p) * (in(q + p) + sign * in(q - p))  (+ the centre column as before),
// and ONE addition per pair serves all NK kernels of the battery: 4 + 4 NK vector operations per step of 2 * 4 * NK multiply-adds
// instead of 8 NK -- 1.7x fewer for NK = 6.  Layout as k_conv_battery: a lane owns one output column and CV_ROWS rows, the weight
// rows slide through scalar registers.  Kernel column kx < r is paired with column 2r - kx: for the input row j of column A = lx + kx
// the partner of output row i (weight row t = j - i) is row 2r - j + 2i of column B = lx + 2r - kx -- rows of one parity, so two
// circular windows of four values (even / odd steps) hold them with ONE new LDS read per step; the step loop is unrolled by 8.
constexpr int CVS_UNROLL = 8;
__host__ __device__ static inline int conv_sym_padded_rows(int radius) { return ((2 * radius + 1 + CV_ROWS - 1 + CVS_UNROLL - 1) / CVS_UNROLL) * CVS_UNROLL; }

template <int NK>
__global__ void __launch_bounds__(256)
k_conv_battery_sym(const double *__restrict__ planes, int H, int W, const double *__restrict__ wgt, int radius, double sign,
                   double clip, double *__restrict__ resp)
{
    extern __shared__ double tile[];                 // [(CV_TY - CV_ROWS + Spad)][(CV_TX + 2r)]
    const int Spad = conv_sym_padded_rows(radius);
    const int tw = CV_TX + 2 * radius, th = CV_TY + 2 * radius, th_pad = CV_TY - CV_ROWS + Spad;
    const int ch = blockIdx.z;
    const double *src = planes + (size_t)ch * H * W;
    const int x0 = blockIdx.x * CV_TX, y0 = blockIdx.y * CV_TY;
    for (int i = threadIdx.x; i < tw * th_pad; i += 256) {
        int ty = i / tw, tx = i - ty * tw;
        // (rows behind the halo only ever meet zero weights: any finite value will do)
        int gy = reflect_index(y0 + min(ty, th - 1) - radius, H), gx = reflect_index(x0 + tx - radius, W);
        tile[i] = src[(size_t)gy * W + gx];
    }
    __syncthreads();
    const int lx = threadIdx.x & 63;
    const int ly = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) * CV_ROWS;
    double acc[NK][CV_ROWS];
#pragma unroll
    for (int k = 0; k < NK; ++k)
#pragma unroll
        for (int i = 0; i < CV_ROWS; ++i) acc[k][i] = 0.0;
    for (int kx = 0; kx < radius; ++kx) {
        const double *wk = wgt + (size_t)kx * Spad * NK;
        const double *colA = tile + lx + kx, *colB = tile + lx + 2 * radius - kx;
        double w[CV_ROWS][NK];
#pragma unroll
        for (int u = 0; u < CV_ROWS; ++u)
#pragma unroll
            for (int k = 0; k < NK; ++k) w[u][k] = 0.0;
        // partner rows of column B: window E serves the even steps (rows 2r - j + 2i, j even), O the odd ones; slot e mod 4
        // takes the row read at even step number e, the output row i looks i even steps back.  Before the first step the windows
        // hold the rows above 2r that meet non-zero weight rows (2r + 2 | 2r + 1, 2r + 3); every other start value only meets
        // the zero rows of the padded weight table
        double E[4] = { 0.0, 0.0, 0.0, colB[(size_t)(ly + 2 * radius + 2) * tw] };
        double O[4] = { 0.0, 0.0, colB[(size_t)(ly + 2 * radius + 3) * tw], colB[(size_t)(ly + 2 * radius + 1) * tw] };
        for (int g = 0; g < Spad; g += CVS_UNROLL) {
#pragma unroll
            for (int u = 0; u < CVS_UNROLL; ++u) {
                const int j = g + u;
#pragma unroll
                for (int k = 0; k < NK; ++k) w[u % CV_ROWS][k] = wk[j * NK + k];         // weight row t = j into slot j mod CV_ROWS
                const double a = colA[(size_t)(ly + j) * tw];
                const double b = colB[(size_t)max(ly + 2 * radius - j, 0) * tw];
                double sv[CV_ROWS];
                if ((u & 1) == 0) {
                    E[(u / 2) % 4] = b;
#pragma unroll
                    for (int i = 0; i < CV_ROWS; ++i) sv[i] = fma(sign, E[((u / 2) - i + 4) % 4], a);
                } else {
                    O[(u / 2) % 4] = b;
#pragma unroll
                    for (int i = 0; i < CV_ROWS; ++i) sv[i] = fma(sign, O[((u / 2) - i + 4) % 4], a);
                }
#pragma unroll
                for (int i = 0; i < CV_ROWS; ++i)
#pragma unroll
                    for (int k = 0; k < NK; ++k) acc[k][i] = fma(w[(u - i + 2 * CV_ROWS) % CV_ROWS][k], sv[i], acc[k][i]);
            }
        }
    }
    {
        // the centre column pairs with itself: the plain sum of k_conv_battery over its Spad rows
        const double *wk = wgt + (size_t)radius * Spad * NK;
        const double *col = tile + lx + radius;
        double w[CV_ROWS][NK];
#pragma unroll
        for (int u = 0; u < CV_ROWS; ++u)
#pragma unroll
            for (int k = 0; k < NK; ++k) w[u][k] = 0.0;
        for (int g = 0; g < Spad; g += CV_ROWS) {
#pragma unroll
            for (int u = 0; u < CV_ROWS; ++u) {
                const int j = g + u;
#pragma unroll
                for (int k = 0; k < NK; ++k) w[u][k] = wk[j * NK + k];
                const double v = col[(size_t)(ly + j) * tw];
#pragma unroll
                for (int i = 0; i < CV_ROWS; ++i)
#pragma unroll
                    for (int k = 0; k < NK; ++k) acc[k][i] = fma(w[(u - i + CV_ROWS) % CV_ROWS][k], v, acc[k][i]);
            }
        }
    }
    const int x = x0 + lx;
#pragma unroll
    for (int i = 0; i < CV_ROWS; ++i) {
        const int y = y0 + ly + i;
        if (x >= W || y >= H) continue;
        double r = acc[0][i];
#pragma unroll
        for (int k = 1; k < NK; ++k) r = fmax(r, acc[k][i]);
        if (r > clip) r = clip;
        resp[(size_t)ch * H * W + (size_t)y * W + x] = r;
    }
}

// ---- dense batteries whose kernels are mirror images of each other in pairs ------------------------------------------------------
// The orientations theta and pi - theta of an edge / bar battery are mirror images: Kb(dy, dx) = m * Ka(dy, -dx) (m = +-1; to the
// last bits of the rotated grid -- the host checks it to 1e-12 of the kernel's maximum and takes Ka's values for both).  Together
// with the point symmetry K(-p) = sign * K(p), the four inputs of a quad (+-dy, +-dx) around the output,
//     top = row c - y, bot = row c + y, A = column cx - x, B = column cx + x,
// enter the two responses only through  U = (A + B)[top] + sign * (A + B)[bot]  and  V = (B - A)[top] - sign * (B - A)[bot]:
//     Ra = WS * U + WD * V,    Rb = m * (WS * U - WD * V),    WS = (Ka(-y, x) + Ka(-y, -x)) / 2,  WD = (Ka(-y, x) - Ka(-y, -x)) / 2,
// i.e. 2 additions + 2 multiply-adds per quad and PAIR of kernels where the point symmetry alone needs 2 + 4 (and the plain sum 8):
// 652 vector operations per column pair, lane and 4 output rows for the three pairs of a Leung-Malik battery (544 multiply-adds
// and U / V additions, 72 column sums / differences, 36 address steps) instead of 1120: 5.84 * 10^8 against 1.07 * 10^9 vector
// instructions per launch at 2048^2 (profiles/rocprof_r04_cfg3_sq_counters.txt), 1.18 against 2.05 ms.
// (The sums differ from the reference's order of additions in the last bits -- as every dense sum here does; the descriptors
// stay 10^-9 from the reference run's, tolerance 10^-5.)  Table wq: [x = 0..R][t = 0..R][WS of the NP pairs | WD of the NP
// pairs], t = R + dy the kernel row (the row of the output itself, t = R, halved by the host: its top and bottom coincide),
// then the NP mirror signs m.  Layout as k_conv_battery: a lane owns one output column and CV_ROWS rows; everything is unrolled
// over the rows (R is a compile-time constant), so the (row, output) combinations that would meet weight rows outside the
// kernel are not computed at all and the weights need no padding.
template <int NP, int R, bool CENTRE>
__device__ __forceinline__ void quad_column(const double *colA, const double *colB, const double *__restrict__ w, double sign, double nsign,
                                            double (&accS)[NP][CV_ROWS], double (&accD)[NP][CV_ROWS])
{
    constexpr int TW = CV_TX + 2 * R;
    double sb[2 * R + CV_ROWS], db[2 * R + CV_ROWS];              // (A + B), (B - A) of the rows below: compile-time indices only
#pragma unroll
    for (int k = 1; k < CV_ROWS; ++k) {
        const double a = colA[(2 * R + k) * TW];
        if (CENTRE) {
            sb[2 * R + k] = a;
        } else {
            const double b = colB[(2 * R + k) * TW];
            sb[2 * R + k] = a + b;
            db[2 * R + k] = b - a;
        }
    }
#pragma unroll
    for (int j = 0; j < R + CV_ROWS; ++j) {
        double st, dt = 0.0;
        {
            const double a = colA[j * TW];
            if (CENTRE) {
                st = a;
            } else {
                const double b = colB[j * TW];
                st = a + b;
                dt = b - a;
            }
        }
        if (j <= R) {
            const double a = colA[(2 * R - j) * TW];
            if (CENTRE) {
                sb[2 * R - j] = a;
            } else {
                const double b = colB[(2 * R - j) * TW];
                sb[2 * R - j] = a + b;
                db[2 * R - j] = b - a;
            }
        }
#pragma unroll
        for (int i = 0; i < CV_ROWS; ++i) {
            const int t = j - i;
            if (t < 0 || t > R) continue;
            const int rho = 2 * R - j + 2 * i;
            const double U = fma(sign, sb[rho], st);
#pragma unroll
            for (int k = 0; k < NP; ++k) accS[k][i] = fma(w[t * 2 * NP + k], U, accS[k][i]);
            if (!CENTRE) {
                const double V = fma(nsign, db[rho], dt);
#pragma unroll
                for (int k = 0; k < NP; ++k) accD[k][i] = fma(w[t * 2 * NP + NP + k], V, accD[k][i]);
            }
        }
    }
}

template <int NP, int R>
__global__ void __launch_bounds__(256)
k_conv_battery_quad(const double *__restrict__ planes, int H, int W, const double *__restrict__ wq, double sign, double clip,
                    double *__restrict__ resp)
{
    extern __shared__ double tile[];                 // [CV_TY + 2R][CV_TX + 2R]
    constexpr int TW = CV_TX + 2 * R, TH = CV_TY + 2 * R;
    const int ch = blockIdx.z;
    const double *src = planes + (size_t)ch * H * W;
    const int x0 = blockIdx.x * CV_TX, y0 = blockIdx.y * CV_TY;
    for (int i = threadIdx.x; i < TW * TH; i += 256) {
        const int ty = i / TW, tx = i - ty * TW;
        const int gy = reflect_index(y0 + ty - R, H), gx = reflect_index(x0 + tx - R, W);
        tile[i] = src[(size_t)gy * W + gx];
    }
    __syncthreads();
    const int lx = threadIdx.x & 63;
    const int ly = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) * CV_ROWS;
    double accS[NP][CV_ROWS], accD[NP][CV_ROWS];
#pragma unroll
    for (int k = 0; k < NP; ++k)
#pragma unroll
        for (int i = 0; i < CV_ROWS; ++i) accS[k][i] = accD[k][i] = 0.0;
    const double nsign = -sign;
    const double *centre = tile + (size_t)ly * TW + lx + R;
    quad_column<NP, R, true>(centre, centre, wq, sign, nsign, accS, accD);
    for (int x = 1; x <= R; ++x)
        quad_column<NP, R, false>(centre - x, centre + x, wq + (size_t)x * (R + 1) * 2 * NP, sign, nsign, accS, accD);
    const double *mirror = wq + (size_t)(R + 1) * (R + 1) * 2 * NP;
    const int x = x0 + lx;
#pragma unroll
    for (int i = 0; i < CV_ROWS; ++i) {
        const int y = y0 + ly + i;
        if (x >= W || y >= H) continue;
        double r = -INFINITY;
#pragma unroll
        for (int k = 0; k < NP; ++k) {
            r = fmax(r, accS[k][i] + accD[k][i]);
            r = fmax(r, mirror[k] * (accS[k][i] - accD[k][i]));
        }
        if (r > clip) r = clip;
        resp[(size_t)ch * H * W + (size_t)y * W + x] = r;
    }
}

// ---- kernels of the bank that are separable -------------------------------------------------------------------------------
// 28 of the 76 Leung-Malik kernels have rank 1 or 2 as 33 x 33 matrices: the Gaussians (rank 1), both Laplacians of a Gaussian
// (rank 2: g''(x) g(y) + g(x) g''(y)) and the edge / bar filters at 0 and 90 degrees (rank 1: gx(3 sigma) gy'(sigma) on the
// unrotated grid) -- 37 % of the 2 * 10^12 flops of an image spent on dense 33 x 33 sums that two 33-tap passes give as well.
// The host factorises every kernel (SVD, pyimsegm_amd._hip); a component is a pair (x taps, y taps) of the flipped kernel, a
// group the components that add up to ONE kernel's response, and the battery's response the maximum over its groups -- and, with
// `merge`, over what the dense kernel has already written for the battery's other orientations.  Per 64 x 16 outputs: the input
// tile with its halo in LDS (as the dense kernel), per component the x pass over all tile rows into a second LDS plane, then the
// y pass from there: 4 * 33 FMAs per output and component instead of 1089 per kernel.
// ST: the kernel side 2 * radius + 1 at compile time (33 for the Leung-Malik bank: both tap loops unroll and the taps of a
// component become scalar registers loaded once, so that an FMA costs one LDS read), or 0: any radius, loops at run time.
constexpr int SEP_MAX_GROUPS = 2;

// what the kernels get of a SepJobs: the taps as offsets from ONE pointer the kernel takes as `const double *__restrict__` -- a
// pointer out of the job table could alias the responses the kernel writes, and the taps would then be fetched by vector loads
// into vector registers instead of scalar loads
struct SepJobDev {
    double *resp;
    long taps_off;
    int groups, rank, merge;
};
struct SepJobsDev {
    int n;
    double *ssq_partial;       // or null; [job][workgroup]: the sum of the squared final responses a workgroup has written
    SepJobDev job[SEP_MAX_JOBS];
};

// One launch serves the separable kernels of up to SEP_MAX_JOBS batteries (the five batteries of one sigma of the bank): the input
// tile -- 4.5 x the plane per launch, the larger part of a launch that only does a rank-1 kernel -- is loaded once for all of them.
template <int ST>
__global__ void __launch_bounds__(256)
k_sep_battery(const double *__restrict__ planes, const double *__restrict__ taps_base, int H, int W, int radius, double clip, SepJobsDev jobs)
{
    extern __shared__ double sep_sm[];
    const int S = ST > 0 ? ST : 2 * radius + 1;
    // (row strides of an odd number of doubles: in the x pass a lane owns a tile ROW, and rows an even number of doubles apart
    // would all start in the same LDS bank)
    const int tw = (CV_TX + 2 * radius) | 1, th = CV_TY + 2 * radius;
    constexpr int TS = CV_TX + 1;
    double *tile = sep_sm;                       // [th][tw]   input
    double *T = sep_sm + (size_t)th * tw;        // [th][TS]   x pass of the current component
    const int ch = blockIdx.z;
    const double *src = planes + (size_t)ch * H * W;
    const int x0 = blockIdx.x * CV_TX, y0 = blockIdx.y * CV_TY;
    for (int i = threadIdx.x; i < tw * th; i += 256) {
        const int ty = i / tw, tx = i - ty * tw;
        tile[i] = src[(size_t)reflect_index(y0 + ty - radius, H) * W + reflect_index(x0 + tx - radius, W)];
    }
    __syncthreads();
    const int lx = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int ly = wave * CV_ROWS;
    const int x = x0 + lx;
    for (int jb = 0; jb < jobs.n; ++jb) {
    const double *__restrict__ taps = taps_base + jobs.job[jb].taps_off;
    double *resp = jobs.job[jb].resp;
    const int n_groups = jobs.job[jb].groups, rank = jobs.job[jb].rank, merge = jobs.job[jb].merge;
    double acc[SEP_MAX_GROUPS][CV_ROWS];
#pragma unroll
    for (int g = 0; g < SEP_MAX_GROUPS; ++g)
#pragma unroll
        for (int i = 0; i < CV_ROWS; ++i) acc[g][i] = 0.0;
#pragma unroll
    for (int g = 0; g < SEP_MAX_GROUPS; ++g) {
        if (g >= n_groups) break;
        for (int c = 0; c < rank; ++c) {
            const double *vx = taps + (size_t)(g * rank + c) * 2 * S, *uy = vx + S;       // wave uniform: scalar loads
            if (ST > 0) {
                // x pass with the lane on a tile ROW and a sliding window along it: four adjacent outputs from 36 LDS reads (the
                // lane-per-column form needs 33 reads per output and is bound by the LDS pipe: 230 us per component and launch at
                // 2048^2, measured); wave w takes the output columns 16 w .. 16 w + 15
                if (lx < th) {
                    const double *rp = tile + (size_t)lx * tw + 16 * wave;
                    double *tp = T + (size_t)lx * TS + 16 * wave;
                    for (int c0 = 0; c0 < 16; c0 += 4) {
                        double o0 = 0.0, o1 = 0.0, o2 = 0.0, o3 = 0.0;
                        // (all reads of the window first, then the arithmetic: with two waves per SIMD nobody hides a read that
                        // is waited for right where it is issued -- 204 000 cycles per wave for 7 600 vector instructions, measured)
                        double win[(ST > 0 ? ST : 1) + 3];
#pragma unroll
                        for (int q = 0; q < (ST > 0 ? ST : 1) + 3; ++q) win[q] = rp[c0 + q];
#pragma unroll
                        for (int q = 0; q < (ST > 0 ? ST : 1) + 3; ++q) {
                            const double v = win[q];
                            if (q < ST) o0 = fma(vx[q < ST ? q : 0], v, o0);
                            if (q >= 1 && q - 1 < ST) o1 = fma(vx[q >= 1 && q - 1 < ST ? q - 1 : 0], v, o1);
                            if (q >= 2 && q - 2 < ST) o2 = fma(vx[q >= 2 && q - 2 < ST ? q - 2 : 0], v, o2);
                            if (q >= 3 && q - 3 < ST) o3 = fma(vx[q >= 3 && q - 3 < ST ? q - 3 : 0], v, o3);
                        }
                        tp[c0] = o0; tp[c0 + 1] = o1; tp[c0 + 2] = o2; tp[c0 + 3] = o3;
                    }
                }
            } else {
                for (int ty = wave; ty < th; ty += 4) {
                    const double *row = tile + (size_t)ty * tw + lx;
                    double v = 0.0;
                    for (int k = 0; k < S; ++k) v = fma(vx[k], row[k], v);
                    T[ty * TS + lx] = v;
                }
            }
            __syncthreads();
            if (ST > 0) {
                // (a value of the x pass serves the up to CV_ROWS output rows it lies in the window of: one LDS read, up to 4 FMAs)
                double col[(ST > 0 ? ST : 1) + CV_ROWS - 1];
#pragma unroll
                for (int q = 0; q < (ST > 0 ? ST : 1) + CV_ROWS - 1; ++q) col[q] = T[(ly + q) * TS + lx];
#pragma unroll
                for (int q = 0; q < (ST > 0 ? ST : 1) + CV_ROWS - 1; ++q) {
                    const double tv = col[q];
#pragma unroll
                    for (int i = 0; i < CV_ROWS; ++i)
                        if (q - i >= 0 && q - i < ST) acc[g][i] = fma(uy[q - i], tv, acc[g][i]);
                }
            } else {
                for (int t = 0; t < S; ++t) {
                    const double w = uy[t];
#pragma unroll
                    for (int i = 0; i < CV_ROWS; ++i) acc[g][i] = fma(w, T[(ly + i + t) * TS + lx], acc[g][i]);
                }
            }
            __syncthreads();
        }
    }
#pragma unroll
    for (int i = 0; i < CV_ROWS; ++i) {
        const int y = y0 + ly + i;
        if (x >= W || y >= H) continue;
        double r = acc[0][i];
        if (n_groups > 1) r = fmax(r, acc[1][i]);
        double *out = resp + (size_t)ch * H * W + (size_t)y * W + x;
        if (merge) r = fmax(r, *out);            // (the dense kernels' maximum, already clipped: min and max commute here)
        if (r > clip) r = clip;
        *out = r;
    }
    }   // jobs
}

// deterministic sum of squares: per-block partials, then one block adds them in a fixed order
__global__ void __launch_bounds__(256) k_sumsq_partial(const double *__restrict__ v, size_t n, double *partial)
{
    __shared__ double sm[256];
    double a = 0.0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) a = fma(v[i], v[i], a);
    sm[threadIdx.x] = a;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) sm[threadIdx.x] += sm[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = sm[0];
}
__global__ void __launch_bounds__(256) k_sumsq_final(const double *partial, int nb, double *out)
{
    __shared__ double sm[256];
    double a = 0.0;
    for (int i = threadIdx.x; i < nb; i += 256) a += partial[i];
    sm[threadIdx.x] = a;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) sm[threadIdx.x] += sm[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = sm[0];
}

int launch_texture_prepare(const void *img, int dtype, int H, int W, const double *taps_dev, int radius,
                           const double *mix_dev, double *planes, double *tmpA, double *tmpB, hipStream_t st, double *fullpad)
{
    const int n = H * W;
    const int grid = cdiv(n, 256);
    if (dtype == DT_U8) hipLaunchKernelGGL(k_to_planes<uint8_t>, grid, 256, 0, st, (const uint8_t *)img, n, planes);
    else if (dtype == DT_F32) hipLaunchKernelGGL(k_to_planes<float>, grid, 256, 0, st, (const float *)img, n, planes);
    else hipLaunchKernelGGL(k_to_planes<double>, grid, 256, 0, st, (const double *)img, n, planes);
    hipLaunchKernelGGL(k_taps_full, cdiv(2 * radius + 1 + 2 * CR, 256), 256, 0, st, taps_dev, radius, fullpad);
    if (launch_long_blur(planes, tmpA, tmpB, 3, H, W, fullpad, radius, st)) return -1;
    hipLaunchKernelGGL(k_mix_subtract, grid, 256, 0, st, planes, tmpB, n, mix_dev, planes);
    HIP_TRY(hipGetLastError());
    return 0;
}

// planes = volume - gaussian_filter(slice, sigma) per slice, P = D planes of H x W
int launch_texture_prepare_volume(const void *vol, int dtype, int P, int H, int W, const double *taps_dev, int radius, double *planes,
                                  double *tmpA, double *tmpB, hipStream_t st, double *fullpad)
{
    const size_t n = (size_t)P * H * W;
    const int grid = cdiv((long)n, 256);
    if (dtype == DT_U8) hipLaunchKernelGGL(k_vol_to_planes<uint8_t>, grid, 256, 0, st, (const uint8_t *)vol, n, planes);
    else if (dtype == DT_F32) hipLaunchKernelGGL(k_vol_to_planes<float>, grid, 256, 0, st, (const float *)vol, n, planes);
    else hipLaunchKernelGGL(k_vol_to_planes<double>, grid, 256, 0, st, (const double *)vol, n, planes);
    hipLaunchKernelGGL(k_taps_full, cdiv(2 * radius + 1 + 2 * CR, 256), 256, 0, st, taps_dev, radius, fullpad);
    if (launch_long_blur(planes, tmpA, tmpB, P, H, W, fullpad, radius, st)) return -1;
    hipLaunchKernelGGL(k_subtract, grid, 256, 0, st, planes, tmpB, n, planes);
    HIP_TRY(hipGetLastError());
    return 0;
}

// ---- the separable kernels of side 33 on a TALL tile ----------------------------------------------------------------------------
// k_sep_battery<33> spends three quarters of its multiply-adds on the x pass over the halo rows of its 64 x 16 tile (48 rows for
// 16 rows of output).  Here a workgroup takes 16 columns x 96 rows: the x pass runs over 128 rows of 16 outputs, the y pass over
// 96 x 16 -- 77 multiply-adds per output and component instead of 165 -- and both passes keep every lane busy:
//   x pass: thread = (tile row, half of its 16 columns): 8 adjacent outputs from a window of 40 values (one LDS read per 6.6 FMAs);
//   y pass: thread = (column, 6 adjacent rows): a window of 38 values of the x pass; 16 lanes write 128 contiguous bytes.
// Sums in the order of k_sep_battery (taps ascending, components in turn): the responses are the same bit for bit.
constexpr int SPT_X = 16, SPT_Y = 96, SPT_R = 16, SPT_S = 2 * SPT_R + 1, SPT_XO = 8, SPT_YO = 6;
constexpr int SPT_TW = (SPT_X + 2 * SPT_R) | 1, SPT_TH = SPT_Y + 2 * SPT_R, SPT_TS = SPT_X + 1;
static_assert(SPT_TH * (SPT_X / SPT_XO) == 256 && SPT_X * (SPT_Y / SPT_YO) == 256, "one item per thread in both passes");

__global__ void __launch_bounds__(256)
k_sep_battery_tall(const double *__restrict__ planes, const double *__restrict__ taps_base, int H, int W, double clip, SepJobsDev jobs)
{
    extern __shared__ double sep_sm[];
    __shared__ double wave_sum[4];
    double *tile = sep_sm;                                   // [SPT_TH][SPT_TW]  input (odd row stride: a lane owns a row in the x pass)
    double *T = sep_sm + (size_t)SPT_TH * SPT_TW;            // [SPT_TH][SPT_TS]  x pass of the current component
    const int ch = blockIdx.z;
    const double *src = planes + (size_t)ch * H * W;
    const int x0 = blockIdx.x * SPT_X, y0 = blockIdx.y * SPT_Y;
    constexpr int IN_W = SPT_X + 2 * SPT_R;
    {   // (a thread's words are requested TOGETHER and stored as they arrive: as `for (i = threadIdx.x; ...; i += 256)` the compiler kept a
        // load, a wait and a store per turn -- a dozen trips to memory one after the other in front of every tile: 754 -> 650 us.
        // The same in k_conv_battery_quad changed nothing -- 1 200 against 1 187 us: four workgroups per CU cover one another's loads)
        constexpr int NL = (IN_W * SPT_TH + 255) / 256;
        double v[NL];
#pragma unroll
        for (int j = 0; j < NL; ++j) {
            const int i = min((int)threadIdx.x + 256 * j, IN_W * SPT_TH - 1);
            const int ty = i / IN_W, tx = i - ty * IN_W;
            v[j] = src[(size_t)reflect_index(y0 + ty - SPT_R, H) * W + reflect_index(x0 + tx - SPT_R, W)];
        }
#pragma unroll
        for (int j = 0; j < NL; ++j) {
            const int i = (int)threadIdx.x + 256 * j;
            const int ty = i / IN_W, tx = i - ty * IN_W;
            if (i < IN_W * SPT_TH) tile[ty * SPT_TW + tx] = v[j];
        }
    }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // x pass: row of the tile, first of its 8 output columns
    const int xrow = (wave & 1) * 64 + lane, xseg = (wave >> 1) * SPT_XO;
    const double *rp = tile + (size_t)xrow * SPT_TW + xseg;
    double *tp = T + (size_t)xrow * SPT_TS + xseg;
    // y pass: column, first of its 6 output rows
    const int ycol = lane & 15, yrow = ((lane >> 4) + 4 * wave) * SPT_YO;
    const double *cp = T + (size_t)yrow * SPT_TS + ycol;
    const int x = x0 + ycol;
    for (int jb = 0; jb < jobs.n; ++jb) {
        const double *__restrict__ taps = taps_base + jobs.job[jb].taps_off;
        double *resp = jobs.job[jb].resp;
        const int n_groups = jobs.job[jb].groups, rank = jobs.job[jb].rank, merge = jobs.job[jb].merge;
        double acc[SEP_MAX_GROUPS][SPT_YO];
#pragma unroll
        for (int g = 0; g < SEP_MAX_GROUPS; ++g)
#pragma unroll
            for (int i = 0; i < SPT_YO; ++i) acc[g][i] = 0.0;
#pragma unroll
        for (int g = 0; g < SEP_MAX_GROUPS; ++g) {
            if (g >= n_groups) break;
            for (int c = 0; c < rank; ++c) {
                const double *vx = taps + (size_t)(g * rank + c) * 2 * SPT_S, *uy = vx + SPT_S;       // wave uniform: scalar loads
                {
                    // (all reads of the window first, then the arithmetic: two waves per SIMD do not hide a read that is waited for
                    // right where it is issued)
                    double win[SPT_S + SPT_XO - 1], o[SPT_XO];
#pragma unroll
                    for (int q = 0; q < SPT_S + SPT_XO - 1; ++q) win[q] = rp[q];
#pragma unroll
                    for (int k = 0; k < SPT_XO; ++k) o[k] = 0.0;
#pragma unroll
                    for (int q = 0; q < SPT_S + SPT_XO - 1; ++q)
#pragma unroll
                        for (int k = 0; k < SPT_XO; ++k)
                            if (q - k >= 0 && q - k < SPT_S) o[k] = fma(vx[q - k], win[q], o[k]);
#pragma unroll
                    for (int k = 0; k < SPT_XO; ++k) tp[k] = o[k];
                }
                __syncthreads();
                {
                    double col[SPT_S + SPT_YO - 1];
#pragma unroll
                    for (int q = 0; q < SPT_S + SPT_YO - 1; ++q) col[q] = cp[q * SPT_TS];
#pragma unroll
                    for (int q = 0; q < SPT_S + SPT_YO - 1; ++q)
#pragma unroll
                        for (int i = 0; i < SPT_YO; ++i)
                            if (q - i >= 0 && q - i < SPT_S) acc[g][i] = fma(uy[q - i], col[q], acc[g][i]);
                }
                __syncthreads();
            }
        }
        double sq = 0.0;
#pragma unroll
        for (int i = 0; i < SPT_YO; ++i) {
            const int y = y0 + yrow + i;
            if (x >= W || y >= H) continue;
            double r = acc[0][i];
            if (n_groups > 1) r = fmax(r, acc[1][i]);
            double *out = resp + (size_t)ch * H * W + (size_t)y * W + x;
            if (merge) r = fmax(r, *out);            // (the dense kernels' maximum, already clipped: min and max commute here)
            if (r > clip) r = clip;
            *out = r;
            sq = fma(r, r, sq);
        }
        if (jobs.ssq_partial) {
            // the battery's response is final here: its sum of squares (the L2 norm of descriptors.py:1090) per workgroup, in a
            // fixed order -- the 100 MB of a battery are not read again for it
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) sq += __shfl_down(sq, off, 64);
            if (lane == 0) wave_sum[wave] = sq;
            __syncthreads();
            if (threadIdx.x == 0) {
                const size_t wg = ((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
                jobs.ssq_partial[(size_t)jb * gridDim.x * gridDim.y * gridDim.z + wg] = ((wave_sum[0] + wave_sum[1]) + wave_sum[2]) + wave_sum[3];
            }
        }
    }
}

// sums of the per-workgroup partial sums of k_sep_battery_tall: block j adds the `count` values of job j in a fixed order
struct SsqTargets {
    double *out[SEP_MAX_JOBS];
};
__global__ void __launch_bounds__(256) k_sumsq_jobs(const double *__restrict__ partial, int count, SsqTargets targets)
{
    __shared__ double sm[256];
    const double *p = partial + (size_t)blockIdx.x * count;
    double a = 0.0;
    for (int i = threadIdx.x; i < count; i += 256) a += p[i];
    sm[threadIdx.x] = a;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) sm[threadIdx.x] += sm[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) targets.out[blockIdx.x][0] = sm[0];
}

// the dense kernels of a battery -> resp (clipped maximum over them); nk = 0: nothing to do
int launch_battery_dense(const double *planes, int H, int W, const double *wgt_dev, int nk, int radius, double clip, double *resp,
                         hipStream_t st, int P, int parity)
{
    if (nk != 0 && nk != 1 && nk != 2 && nk != 4 && nk != 6 && nk != 8) {
        set_error("filter battery: 1, 2, 4, 6 or 8 dense kernels per battery are supported");
        return -1;
    }
    if (nk == 0) return 0;
    const int S = 2 * radius + 1, Spad = conv_padded_rows(radius);
    size_t lds = (size_t)(CV_TX + 2 * radius) * (CV_TY - CV_ROWS + Spad) * sizeof(double);
    if (lds > 150 * 1024) {
        set_error("filter battery: kernel radius too large for the LDS tile");
        return -1;
    }
    dim3 grid(cdiv(W, CV_TX), cdiv(H, CV_TY), P);
    if (parity == 2 || parity == -2) {
        // point symmetry AND mirror pairs: wgt_dev holds the quad table of k_conv_battery_quad (nk / 2 pairs)
        constexpr int R = 16;
        if (radius != R || (nk & 1)) {
            set_error("filter battery: the mirror-pair form takes kernels of side 33 in pairs");
            return -1;
        }
        const size_t ldsq = (size_t)(CV_TX + 2 * R) * (CV_TY + 2 * R) * sizeof(double);
        const double sign = parity > 0 ? 1.0 : -1.0;
#define LAUNCH_QUAD(NP)                                                                                                          \
    {                                                                                                                            \
        HIP_TRY(hipFuncSetAttribute((const void *)k_conv_battery_quad<NP, R>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsq)); \
        hipLaunchKernelGGL((k_conv_battery_quad<NP, R>), grid, 256, ldsq, st, planes, H, W, wgt_dev, sign, clip, resp);          \
    }
        if (nk == 2) LAUNCH_QUAD(1) else if (nk == 4) LAUNCH_QUAD(2) else if (nk == 6) LAUNCH_QUAD(3) else LAUNCH_QUAD(4)
#undef LAUNCH_QUAD
    } else if (parity != 0) {
        // every dense kernel of the battery is even (+1) or odd (-1) under the point reflection: half the multiplications
        const int Spad8 = conv_sym_padded_rows(radius);
        const size_t lds8 = (size_t)(CV_TX + 2 * radius) * (CV_TY - CV_ROWS + Spad8) * sizeof(double);
        if (lds8 > 150 * 1024) {
            set_error("filter battery: kernel radius too large for the LDS tile");
            return -1;
        }
        double *wpad = const_cast<double *>(wgt_dev) + (size_t)S * S * nk;
        hipLaunchKernelGGL(k_pad_weights, cdiv((long)S * Spad8 * nk, 256), 256, 0, st, wgt_dev, S, Spad8, nk, wpad);
        const void *fn = nk == 8 ? (const void *)k_conv_battery_sym<8> : nk == 6 ? (const void *)k_conv_battery_sym<6>
                       : nk == 4 ? (const void *)k_conv_battery_sym<4> : nk == 2 ? (const void *)k_conv_battery_sym<2>
                                                                                 : (const void *)k_conv_battery_sym<1>;
        if (lds8 > 48 * 1024) HIP_TRY(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds8));
        const double sign = parity > 0 ? 1.0 : -1.0;
        if (nk == 8) hipLaunchKernelGGL(k_conv_battery_sym<8>, grid, 256, lds8, st, planes, H, W, wpad, radius, sign, clip, resp);
        else if (nk == 6) hipLaunchKernelGGL(k_conv_battery_sym<6>, grid, 256, lds8, st, planes, H, W, wpad, radius, sign, clip, resp);
        else if (nk == 4) hipLaunchKernelGGL(k_conv_battery_sym<4>, grid, 256, lds8, st, planes, H, W, wpad, radius, sign, clip, resp);
        else if (nk == 2) hipLaunchKernelGGL(k_conv_battery_sym<2>, grid, 256, lds8, st, planes, H, W, wpad, radius, sign, clip, resp);
        else hipLaunchKernelGGL(k_conv_battery_sym<1>, grid, 256, lds8, st, planes, H, W, wpad, radius, sign, clip, resp);
    } else {
        // (the padded table lives behind the caller's weights: the caller reserves S * (S + 16) * nk doubles there)
        double *wpad = const_cast<double *>(wgt_dev) + (size_t)S * S * nk;
        hipLaunchKernelGGL(k_pad_weights, cdiv((long)S * Spad * nk, 256), 256, 0, st, wgt_dev, S, Spad, nk, wpad);
        const void *fn = nk == 8 ? (const void *)k_conv_battery<8> : nk == 6 ? (const void *)k_conv_battery<6>
                       : nk == 4 ? (const void *)k_conv_battery<4> : nk == 2 ? (const void *)k_conv_battery<2> : (const void *)k_conv_battery<1>;
        if (lds > 48 * 1024) HIP_TRY(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        if (nk == 8) hipLaunchKernelGGL(k_conv_battery<8>, grid, 256, lds, st, planes, H, W, wpad, radius, clip, resp);
        else if (nk == 6) hipLaunchKernelGGL(k_conv_battery<6>, grid, 256, lds, st, planes, H, W, wpad, radius, clip, resp);
        else if (nk == 4) hipLaunchKernelGGL(k_conv_battery<4>, grid, 256, lds, st, planes, H, W, wpad, radius, clip, resp);
        else if (nk == 2) hipLaunchKernelGGL(k_conv_battery<2>, grid, 256, lds, st, planes, H, W, wpad, radius, clip, resp);
        else hipLaunchKernelGGL(k_conv_battery<1>, grid, 256, lds, st, planes, H, W, wpad, radius, clip, resp);
    }
    HIP_TRY(hipGetLastError());
    return 0;
}

// the separable kernels of up to SEP_MAX_JOBS batteries in one launch (one load of the input tile for all of them)
// doubles of scratch launch_battery_sep needs to form the sums of squares of `jobs` batteries itself (0: it does not, for this
// kernel size: launch_response_sumsq does it)
size_t sep_sumsq_scratch(int H, int W, int P, int radius, int jobs)
{
    if (radius != SPT_R || knobs().sep_wide_tile) return 0;
    return (size_t)jobs * cdiv(W, SPT_X) * cdiv(H, SPT_Y) * P;
}

// `ssq_scratch` (sep_sumsq_scratch doubles) with `ssq_out[j]` per job: the sum of the squared responses of job j is formed on the
// way (every job is then the LAST writer of its response); null: not
int launch_battery_sep(const double *planes, int H, int W, int radius, double clip, const SepJobs &jobs, hipStream_t st, int P,
                       double *ssq_scratch, double *const *ssq_out)
{
    if (jobs.n < 1) return 0;
    for (int j = 0; j < jobs.n; ++j) {
        const SepJob &q = jobs.job[j];
        if (jobs.n > SEP_MAX_JOBS || q.groups < 1 || q.groups > SEP_MAX_GROUPS || !q.taps || !q.resp || q.rank < 1 || q.rank > 4) {
            set_error("filter battery: up to 2 separable kernels of rank 1..4 per battery, up to 5 batteries per launch");
            return -1;
        }
    }
    const size_t sep_lds = ((size_t)((CV_TX + 2 * radius) | 1) + CV_TX + 1) * (CV_TY + 2 * radius) * sizeof(double);
    if (sep_lds > 150 * 1024) {
        set_error("filter battery: kernel radius too large for the LDS tile");
        return -1;
    }
    SepJobsDev dev = {};
    dev.n = jobs.n;
    const double *taps_base = jobs.job[0].taps;
    for (int j = 0; j < jobs.n; ++j) {
        const SepJob &q = jobs.job[j];
        dev.job[j].resp = q.resp; dev.job[j].taps_off = (long)(q.taps - taps_base);          // (the taps of a call lie in one buffer)
        dev.job[j].groups = q.groups; dev.job[j].rank = q.rank; dev.job[j].merge = q.merge;
    }
    if (radius == SPT_R && !knobs().sep_wide_tile) {
        const size_t lds = (size_t)SPT_TH * (SPT_TW + SPT_TS) * sizeof(double);
        HIP_TRY(hipFuncSetAttribute((const void *)k_sep_battery_tall, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        dev.ssq_partial = ssq_out ? ssq_scratch : nullptr;
        const dim3 grid(cdiv(W, SPT_X), cdiv(H, SPT_Y), P);
        hipLaunchKernelGGL(k_sep_battery_tall, grid, 256, lds, st, planes, taps_base, H, W, clip, dev);
        if (ssq_out) {
            SsqTargets targets = {};
            for (int j = 0; j < jobs.n; ++j) targets.out[j] = ssq_out[j];
            hipLaunchKernelGGL(k_sumsq_jobs, jobs.n, 256, 0, st, ssq_scratch, (int)(grid.x * grid.y * grid.z), targets);
        }
        HIP_TRY(hipGetLastError());
        return 0;
    }
    if (ssq_out) {
        set_error("filter battery: sums of squares with the separable kernels only for kernels of side 33");
        return -1;
    }
    dim3 grid(cdiv(W, CV_TX), cdiv(H, CV_TY), P);
    const void *sfn = radius == 16 ? (const void *)k_sep_battery<33> : (const void *)k_sep_battery<0>;
    if (sep_lds > 48 * 1024) HIP_TRY(hipFuncSetAttribute(sfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sep_lds));
    if (radius == 16) hipLaunchKernelGGL(k_sep_battery<33>, grid, 256, sep_lds, st, planes, taps_base, H, W, radius, clip, dev);
    else hipLaunchKernelGGL(k_sep_battery<0>, grid, 256, sep_lds, st, planes, taps_base, H, W, radius, clip, dev);
    HIP_TRY(hipGetLastError());
    return 0;
}

// sum of squares of a response (P planes) -> sumsq_dev[0], deterministic
int launch_response_sumsq(const double *resp, size_t count, double *partial, double *sumsq_dev, hipStream_t st)
{
    const int nb = 1024;
    hipLaunchKernelGGL(k_sumsq_partial, nb, 256, 0, st, resp, count, partial);
    hipLaunchKernelGGL(k_sumsq_final, 1, 256, 0, st, partial, nb, sumsq_dev);
    HIP_TRY(hipGetLastError());
    return 0;
}

int launch_filter_battery(const double *planes, int H, int W, const double *wgt_dev, int nk, int radius, double clip,
                          double *resp, double *partial, double *sumsq_dev, hipStream_t st, int P, const double *sep_dev, int sep_groups,
                          int sep_rank, int parity)
{
    if (nk == 0 && sep_groups == 0) {
        set_error("filter battery: at least one kernel");
        return -1;
    }
    if (launch_battery_dense(planes, H, W, wgt_dev, nk, radius, clip, resp, st, P, parity)) return -1;
    if (sep_groups > 0) {
        SepJobs jobs;
        jobs.n = 1;
        jobs.job[0].resp = resp; jobs.job[0].taps = sep_dev; jobs.job[0].groups = sep_groups; jobs.job[0].rank = sep_rank;
        jobs.job[0].merge = nk > 0 ? 1 : 0;
        if (launch_battery_sep(planes, H, W, radius, clip, jobs, st, P, nullptr, nullptr)) return -1;
    }
    return launch_response_sumsq(resp, (size_t)P * H * W, partial, sumsq_dev, st);
}

}  // namespace imsegm
