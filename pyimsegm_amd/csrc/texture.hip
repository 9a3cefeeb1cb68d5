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

// symmetric 1-D correlation along y (AXIS 0) or x (AXIS 1) with a long kernel (sigma = 150 ->
// radius 600); one output per lane, lanes run along x so every tap is a coalesced row segment
template <int AXIS>
__global__ void __launch_bounds__(256)
k_corr1d_long(const double *__restrict__ src, double *__restrict__ dst, int H, int W, const double *__restrict__ taps,
              int radius)
{
    int x = blockIdx.x * 64 + (threadIdx.x & 63);
    int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= W || y >= H) return;
    const double *s = src + (size_t)blockIdx.z * H * W;
    double acc = s[(size_t)y * W + x] * taps[0];
    if (AXIS == 0) {
        const bool interior = y - radius >= 0 && y + radius < H;
        for (int j = 1; j <= radius; ++j) {
            int ya = interior ? y - j : reflect_index(y - j, H), yb = interior ? y + j : reflect_index(y + j, H);
            acc = fma(s[(size_t)ya * W + x] + s[(size_t)yb * W + x], taps[j], acc);
        }
    } else {
        for (int j = 1; j <= radius; ++j) {
            int xa = reflect_index(x - j, W), xb = reflect_index(x + j, W);
            acc = fma(s[(size_t)y * W + xa] + s[(size_t)y * W + xb], taps[j], acc);
        }
    }
    dst[(size_t)blockIdx.z * H * W + (size_t)y * W + x] = acc;
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

template <int NK>
__global__ void __launch_bounds__(256)
k_conv_battery(const double *__restrict__ planes, int H, int W, const double *__restrict__ wgt, int radius,
               double clip, double *__restrict__ resp)
{
    extern __shared__ double tile[];                 // [(CV_TY + 2r)][(CV_TX + 2r)]
    const int S = 2 * radius + 1;
    const int tw = CV_TX + 2 * radius, th = CV_TY + 2 * radius;
    const int ch = blockIdx.z;
    const double *src = planes + (size_t)ch * H * W;
    const int x0 = blockIdx.x * CV_TX, y0 = blockIdx.y * CV_TY;
    for (int i = threadIdx.x; i < tw * th; i += 256) {
        int ty = i / tw, tx = i - ty * tw;
        int gy = reflect_index(y0 + ty - radius, H), gx = reflect_index(x0 + tx - radius, W);
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
    // correlation form: out[y][x] = sum_{t, kx} Wc[t][kx] * in[y + t - r][x + kx - r]; the host passes the
    // flipped kernels, so this equals ndimage.convolve
    for (int kx = 0; kx < S; ++kx) {
        const double *wk = wgt + (size_t)kx * S * NK;
        for (int ky = 0; ky < S + CV_ROWS - 1; ++ky) {
            const double v = tile[(ly + ky) * tw + lx + kx];
#pragma unroll
            for (int i = 0; i < CV_ROWS; ++i) {
                const int t = ky - i;                   // tap row for output row i (wave-uniform)
                if (t < 0 || t >= S) continue;
#pragma unroll
                for (int k = 0; k < NK; ++k) acc[k][i] = fma(wk[t * NK + k], v, acc[k][i]);
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
                           const double *mix_dev, double *planes, double *tmpA, double *tmpB, hipStream_t st)
{
    const int n = H * W;
    const int grid = cdiv(n, 256);
    if (dtype == DT_U8) hipLaunchKernelGGL(k_to_planes<uint8_t>, grid, 256, 0, st, (const uint8_t *)img, n, planes);
    else if (dtype == DT_F32) hipLaunchKernelGGL(k_to_planes<float>, grid, 256, 0, st, (const float *)img, n, planes);
    else hipLaunchKernelGGL(k_to_planes<double>, grid, 256, 0, st, (const double *)img, n, planes);
    dim3 g(cdiv(W, 64), cdiv(H, 4), 3);
    hipLaunchKernelGGL(k_corr1d_long<0>, g, 256, 0, st, planes, tmpA, H, W, taps_dev, radius);
    hipLaunchKernelGGL(k_corr1d_long<1>, g, 256, 0, st, tmpA, tmpB, H, W, taps_dev, radius);
    hipLaunchKernelGGL(k_mix_subtract, grid, 256, 0, st, planes, tmpB, n, mix_dev, planes);
    HIP_TRY(hipGetLastError());
    return 0;
}

// planes = volume - gaussian_filter(slice, sigma) per slice, P = D planes of H x W
int launch_texture_prepare_volume(const void *vol, int dtype, int P, int H, int W, const double *taps_dev, int radius, double *planes,
                                  double *tmpA, double *tmpB, hipStream_t st)
{
    const size_t n = (size_t)P * H * W;
    const int grid = cdiv((long)n, 256);
    if (dtype == DT_U8) hipLaunchKernelGGL(k_vol_to_planes<uint8_t>, grid, 256, 0, st, (const uint8_t *)vol, n, planes);
    else if (dtype == DT_F32) hipLaunchKernelGGL(k_vol_to_planes<float>, grid, 256, 0, st, (const float *)vol, n, planes);
    else hipLaunchKernelGGL(k_vol_to_planes<double>, grid, 256, 0, st, (const double *)vol, n, planes);
    dim3 g(cdiv(W, 64), cdiv(H, 4), P);
    hipLaunchKernelGGL(k_corr1d_long<0>, g, 256, 0, st, planes, tmpA, H, W, taps_dev, radius);
    hipLaunchKernelGGL(k_corr1d_long<1>, g, 256, 0, st, tmpA, tmpB, H, W, taps_dev, radius);
    hipLaunchKernelGGL(k_subtract, grid, 256, 0, st, planes, tmpB, n, planes);
    HIP_TRY(hipGetLastError());
    return 0;
}

int launch_filter_battery(const double *planes, int H, int W, const double *wgt_dev, int nk, int radius, double clip,
                          double *resp, double *partial, double *sumsq_dev, hipStream_t st, int P)
{
    if (nk != 1 && nk != 2 && nk != 4 && nk != 8) {
        set_error("filter battery: 1, 2, 4 or 8 kernels per battery are supported");
        return -1;
    }
    size_t lds = (size_t)(CV_TX + 2 * radius) * (CV_TY + 2 * radius) * sizeof(double);
    if (lds > 150 * 1024) {
        set_error("filter battery: kernel radius too large for the LDS tile");
        return -1;
    }
    dim3 grid(cdiv(W, CV_TX), cdiv(H, CV_TY), P);
    const void *fn = nk == 8 ? (const void *)k_conv_battery<8> : nk == 4 ? (const void *)k_conv_battery<4>
                   : nk == 2 ? (const void *)k_conv_battery<2> : (const void *)k_conv_battery<1>;
    if (lds > 48 * 1024) HIP_TRY(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    if (nk == 8) hipLaunchKernelGGL(k_conv_battery<8>, grid, 256, lds, st, planes, H, W, wgt_dev, radius, clip, resp);
    else if (nk == 4) hipLaunchKernelGGL(k_conv_battery<4>, grid, 256, lds, st, planes, H, W, wgt_dev, radius, clip, resp);
    else if (nk == 2) hipLaunchKernelGGL(k_conv_battery<2>, grid, 256, lds, st, planes, H, W, wgt_dev, radius, clip, resp);
    else hipLaunchKernelGGL(k_conv_battery<1>, grid, 256, lds, st, planes, H, W, wgt_dev, radius, clip, resp);
    const int nb = 1024;
    hipLaunchKernelGGL(k_sumsq_partial, nb, 256, 0, st, resp, (size_t)P * H * W, partial);
    hipLaunchKernelGGL(k_sumsq_final, 1, 256, 0, st, partial, nb, sumsq_dev);
    HIP_TRY(hipGetLastError());
    return 0;
}

}  // namespace imsegm
