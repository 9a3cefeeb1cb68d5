// output.hip -- output side of the driver: the background swap of `imsegm.labeling.assume_bg_on_boundary`
// (/root/reference/imsegm/labeling.py:719-753 with get_image2d_boundary_color, utilities/data_io.py:1002-1036) and the
// narrow result formats of the fused segmentation call (uint8 class map, float32 soft segmentation) for gfx950.
#include "slic.h"

namespace imsegm {

struct Rects {
    int r0[4], r1[4], c0[4], c1[4];      // the four border strips, as numpy slices them (they may overlap: pixels count per strip)
};

// min / max label over the strips (order-preserving int atomics)
__global__ void __launch_bounds__(256) k_rect_minmax(const int32_t *__restrict__ labels, int W, Rects rc, int32_t *__restrict__ out)
{
    const int q = blockIdx.y;
    const int h = rc.r1[q] - rc.r0[q], w = rc.c1[q] - rc.c0[q];
    const long area = (long)max(h, 0) * max(w, 0);
    int mn = 0x7fffffff, mx = (int)0x80000000;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < area; i += (long)gridDim.x * 256) {
        const int y = rc.r0[q] + (int)(i / w), x = rc.c0[q] + (int)(i % w);
        const int v = labels[(size_t)y * W + x];
        mn = min(mn, v);
        mx = max(mx, v);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        mn = min(mn, __shfl_xor(mn, off, 64));
        mx = max(mx, __shfl_xor(mx, off, 64));
    }
    if ((threadIdx.x & 63) == 0 && mn <= mx) {
        atomicMin(out, mn);
        atomicMax(out + 1, mx);
    }
}

// np.bincount of the strips' pixels: LDS histogram for up to 4096 labels, global atomics beyond
constexpr int HIST_LDS = 4096;
__global__ void __launch_bounds__(256) k_rect_hist(const int32_t *__restrict__ labels, int W, Rects rc, unsigned long long *__restrict__ hist, int nb)
{
    __shared__ unsigned int lh[HIST_LDS];
    const bool lds = nb <= HIST_LDS;
    if (lds) {
        for (int i = threadIdx.x; i < nb; i += 256) lh[i] = 0;
        __syncthreads();
    }
    const int q = blockIdx.y;
    const int h = rc.r1[q] - rc.r0[q], w = rc.c1[q] - rc.c0[q];
    const long area = (long)max(h, 0) * max(w, 0);
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < area; i += (long)gridDim.x * 256) {
        const int y = rc.r0[q] + (int)(i / w), x = rc.c0[q] + (int)(i % w);
        const int v = labels[(size_t)y * W + x];
        if (v < 0 || v >= nb) continue;
        if (lds) atomicAdd(&lh[v], 1u);
        else atomicAdd(&hist[v], 1ull);
    }
    if (lds) {
        __syncthreads();
        for (int i = threadIdx.x; i < nb; i += 256)
            if (lh[i]) atomicAdd(&hist[i], (unsigned long long)lh[i]);
    }
}

// lut = identity with a <-> b exchanged, applied in place (labeling.py:748-752)
__global__ void __launch_bounds__(256) k_swap_labels(int32_t *__restrict__ labels, size_t n, int a, int b)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const int v = labels[i];
        labels[i] = v == a ? b : (v == b ? a : v);
    }
}

int launch_boundary_minmax(const int32_t *labels, int W, const int rects[16], int32_t *out2_dev, hipStream_t st)
{
    Rects rc;
    for (int q = 0; q < 4; ++q) { rc.r0[q] = rects[4 * q]; rc.r1[q] = rects[4 * q + 1]; rc.c0[q] = rects[4 * q + 2]; rc.c1[q] = rects[4 * q + 3]; }
    const int32_t init[2] = { 0x7fffffff, (int32_t)0x80000000 };
    HIP_TRY(hipMemcpyAsync(out2_dev, init, sizeof(init), hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_rect_minmax, dim3(64, 4), 256, 0, st, labels, W, rc, out2_dev);
    HIP_TRY(hipGetLastError());
    return 0;
}

int launch_boundary_hist(const int32_t *labels, int W, const int rects[16], unsigned long long *hist_dev, int nb, hipStream_t st)
{
    Rects rc;
    for (int q = 0; q < 4; ++q) { rc.r0[q] = rects[4 * q]; rc.r1[q] = rects[4 * q + 1]; rc.c0[q] = rects[4 * q + 2]; rc.c1[q] = rects[4 * q + 3]; }
    HIP_TRY(hipMemsetAsync(hist_dev, 0, (size_t)nb * sizeof(unsigned long long), st));
    hipLaunchKernelGGL(k_rect_hist, dim3(64, 4), 256, 0, st, labels, W, rc, hist_dev, nb);
    HIP_TRY(hipGetLastError());
    return 0;
}

int launch_swap_labels(int32_t *labels, size_t n, int a, int b, hipStream_t st)
{
    hipLaunchKernelGGL(k_swap_labels, (int)std::min<size_t>(cdiv((long)n, 256 * 8), 4096), 256, 0, st, labels, n, a, b);
    HIP_TRY(hipGetLastError());
    return 0;
}

// ---- narrow result formats ------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_narrow_i32_u8(const int32_t *__restrict__ src, uint8_t *__restrict__ dst, size_t n)
{
    // four labels per thread: one 16-byte load, one 4-byte store
    const size_t n4 = n / 4;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const int4 v = reinterpret_cast<const int4 *>(src)[i];
        reinterpret_cast<uint32_t *>(dst)[i] = (uint32_t)(uint8_t)v.x | ((uint32_t)(uint8_t)v.y << 8) | ((uint32_t)(uint8_t)v.z << 16) | ((uint32_t)(uint8_t)v.w << 24);
    }
    if (blockIdx.x == 0)
        for (size_t i = n4 * 4 + threadIdx.x; i < n; i += 256) dst[i] = (uint8_t)src[i];
}
__global__ void __launch_bounds__(256) k_narrow_f64_f32(const double *__restrict__ src, float *__restrict__ dst, size_t n)
{
    const size_t n2 = n / 2;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n2; i += (size_t)gridDim.x * 256) {
        const double2 v = reinterpret_cast<const double2 *>(src)[i];
        reinterpret_cast<float2 *>(dst)[i] = make_float2((float)v.x, (float)v.y);
    }
    if (blockIdx.x == 0 && threadIdx.x == 0 && (n & 1)) dst[n - 1] = (float)src[n - 1];
}

int launch_narrow_labels_u8(const int32_t *src, uint8_t *dst, size_t n, hipStream_t st)
{
    hipLaunchKernelGGL(k_narrow_i32_u8, (int)std::min<size_t>(std::max<size_t>(cdiv((long)n, 1024), 1), 4096), 256, 0, st, src, dst, n);
    HIP_TRY(hipGetLastError());
    return 0;
}
int launch_narrow_soft_f32(const double *src, float *dst, size_t n, hipStream_t st)
{
    hipLaunchKernelGGL(k_narrow_f64_f32, (int)std::min<size_t>(std::max<size_t>(cdiv((long)n, 512), 1), 8192), 256, 0, st, src, dst, n);
    HIP_TRY(hipGetLastError());
    return 0;
}

// ---- NaN / inf in the uploaded pixels? ------------------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256) k_count_nonfinite(const T *__restrict__ src, size_t n, unsigned int *count)
{
    unsigned int bad = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) bad += !isfinite((double)src[i]);
    if (__syncthreads_or(bad) && bad) atomicAdd(count, bad);
}

int launch_count_nonfinite(const void *src, int dtype, size_t n, unsigned int *count_dev, hipStream_t st)
{
    HIP_TRY(hipMemsetAsync(count_dev, 0, sizeof(unsigned int), st));
    const int grid = (int)std::min<size_t>(std::max<size_t>(cdiv((long)n, 2048), 1), 8192);
    if (dtype == DT_F32) hipLaunchKernelGGL(k_count_nonfinite<float>, grid, 256, 0, st, (const float *)src, n, count_dev);
    else if (dtype == DT_F64) hipLaunchKernelGGL(k_count_nonfinite<double>, grid, 256, 0, st, (const double *)src, n, count_dev);
    HIP_TRY(hipGetLastError());
    return 0;
}

}  // namespace imsegm
