// median.hip -- the two remaining per-superpixel statistics of compute_image2d_color_statistic /
// compute_image3d_gray_statistic (/root/reference/imsegm/descriptors.py:705-863):
//   'median'   numpy_img2d_color_median :420-455 / numpy_img3d_gray_median :671-702 (Python lists per label + np.median)
//   'meanGrad' np.sum(np.gradient(channel), axis=0) per 2-D slice, stored in the image's dtype, then the segmented mean
//              :766-770, :841-845
// Median: the pixels of a channel are ordered by (label, value) with two stable radix sorts (hipCUB: value bits first,
// then the label), the per-label counts give the segment offsets, one thread per label picks the middle element(s).
// Exact: no arithmetic but the mean of the two middle values, formed in the image's precision as np.median does.
#include "slic.h"

#include <hipcub/hipcub.hpp>

namespace imsegm {

__device__ __forceinline__ unsigned long long order_key(double v)
{
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    return (b & 0x8000000000000000ULL) ? ~b : (b | 0x8000000000000000ULL);      // total order of IEEE doubles
}
__device__ __forceinline__ double key_value(unsigned long long k)
{
    const unsigned long long b = (k & 0x8000000000000000ULL) ? (k & 0x7fffffffffffffffULL) : ~k;
    return __longlong_as_double((long long)b);
}

template <typename T>
__global__ void __launch_bounds__(256)
k_median_keys(const T *__restrict__ img, int C, int c, size_t n, const int32_t *__restrict__ labels, int K,
              unsigned long long *__restrict__ keys, int32_t *__restrict__ lab_out, unsigned int *__restrict__ counts)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int l = labels[i];
    if (l < 0 || l >= K) l = K;                                   // out-of-range labels sort behind everything
    keys[i] = order_key((double)img[i * C + c]);                  // uint8 / float32 -> double is exact and monotone
    lab_out[i] = l;
    if (counts && l < K) atomicAdd(&counts[l], 1u);
}

// dtype: DT_U8 / DT_F64 -> mean of the two middle values in float64, DT_F32 -> in float32 (np.mean of a float32 pair)
__global__ void __launch_bounds__(256)
k_median_pick(const unsigned long long *__restrict__ keys, const unsigned int *__restrict__ offsets,
              const unsigned int *__restrict__ counts, int K, int C, int c, int dtype, double *__restrict__ out)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= K) return;
    const unsigned int n = counts[k], o = offsets[k];
    double m;
    if (n == 0) {
        m = __longlong_as_double(0x7ff8000000000000LL);           // np.median([]) is nan
    } else if (n & 1u) {
        m = key_value(keys[o + n / 2]);
    } else {
        const double a = key_value(keys[o + n / 2 - 1]), b = key_value(keys[o + n / 2]);
        if (dtype == DT_F32) m = (double)(((float)a + (float)b) / 2.0f);
        else m = (a + b) / 2.0;
    }
    out[(size_t)k * C + c] = m;
}

// gradient image: per slice and channel np.gradient along rows + along columns (second-order central differences in the
// interior, one-sided first differences at the two ends; integer images are differenced in float64, float32 images in
// float32), the sum cast back to the image's dtype -- integers truncate towards zero and wrap, as numpy's cast does
template <typename T> struct GradT { typedef double type; };
template <> struct GradT<float> { typedef float type; };

template <typename T>
__global__ void __launch_bounds__(256)
k_gradient_image(const T *__restrict__ src, T *__restrict__ dst, int S, int H, int W, int C)
{
    typedef typename GradT<T>::type F;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t total = (size_t)S * H * W * C;
    if (i >= total) return;
    const int c = (int)(i % C);
    const size_t px = i / C;
    const int x = (int)(px % W), y = (int)((px / W) % H);
    const size_t row = (size_t)W * C, col = (size_t)C;
    F gy, gx;
    if (y == 0) gy = ((F)src[i + row] - (F)src[i]) / (F)1;
    else if (y == H - 1) gy = ((F)src[i] - (F)src[i - row]) / (F)1;
    else gy = ((F)src[i + row] - (F)src[i - row]) / (F)2;
    if (x == 0) gx = ((F)src[i + col] - (F)src[i]) / (F)1;
    else if (x == W - 1) gx = ((F)src[i] - (F)src[i - col]) / (F)1;
    else gx = ((F)src[i + col] - (F)src[i - col]) / (F)2;
    const F g = gy + gx;
    if (sizeof(T) == 1) dst[i] = (T)(unsigned char)(int)g;        // float64 -> uint8 as the x86 cast: truncate, wrap
    else dst[i] = (T)g;
    (void)c;
}

int launch_gradient_image(const void *src, void *dst, int dtype, int S, int H, int W, int C, hipStream_t st)
{
    const size_t total = (size_t)S * H * W * C;
    const int grid = cdiv((long)total, 256);
    if (dtype == DT_U8) hipLaunchKernelGGL(k_gradient_image<uint8_t>, grid, 256, 0, st, (const uint8_t *)src, (uint8_t *)dst, S, H, W, C);
    else if (dtype == DT_F32) hipLaunchKernelGGL(k_gradient_image<float>, grid, 256, 0, st, (const float *)src, (float *)dst, S, H, W, C);
    else hipLaunchKernelGGL(k_gradient_image<double>, grid, 256, 0, st, (const double *)src, (double *)dst, S, H, W, C);
    HIP_TRY(hipGetLastError());
    return 0;
}

size_t median_scratch_bytes(size_t n, int K)
{
    size_t t1 = 0, t2 = 0, t3 = 0;
    (void)hipcub::DeviceRadixSort::SortPairs(nullptr, t1, (const unsigned long long *)nullptr, (unsigned long long *)nullptr,
                                             (const int32_t *)nullptr, (int32_t *)nullptr, (int)n, 0, 64, (hipStream_t) nullptr);
    (void)hipcub::DeviceRadixSort::SortPairs(nullptr, t2, (const int32_t *)nullptr, (int32_t *)nullptr, (const unsigned long long *)nullptr,
                                             (unsigned long long *)nullptr, (int)n, 0, 32, (hipStream_t) nullptr);
    (void)hipcub::DeviceScan::ExclusiveSum(nullptr, t3, (const unsigned int *)nullptr, (unsigned int *)nullptr, K, (hipStream_t) nullptr);
    const size_t tmp = std::max(t1, std::max(t2, t3));
    return 2 * n * 8 + 2 * n * 4 + 2 * ((size_t)K + 64) * 4 + tmp + 1024;
}

// out: [K][C] medians on the device; scratch: median_scratch_bytes(n, K)
int launch_segment_median(const void *img, int dtype, int C, size_t n, const int32_t *labels, int K, void *scratch, size_t scratch_bytes,
                          double *out, hipStream_t st)
{
    if (n > 0x7fffffffULL) {
        set_error("median: more than 2^31 elements");
        return -1;
    }
    unsigned char *b = static_cast<unsigned char *>(scratch);
    unsigned long long *keyA = reinterpret_cast<unsigned long long *>(b); b += n * 8;
    unsigned long long *keyB = reinterpret_cast<unsigned long long *>(b); b += n * 8;
    int32_t *labA = reinterpret_cast<int32_t *>(b); b += n * 4;
    int32_t *labB = reinterpret_cast<int32_t *>(b); b += n * 4;
    unsigned int *counts = reinterpret_cast<unsigned int *>(b); b += ((size_t)K + 64) * 4;
    unsigned int *offsets = reinterpret_cast<unsigned int *>(b); b += ((size_t)K + 64) * 4;
    b = reinterpret_cast<unsigned char *>(((uintptr_t)b + 255) & ~(uintptr_t)255);
    size_t tmp_bytes = scratch_bytes - (size_t)(b - static_cast<unsigned char *>(scratch));
    int label_bits = 1;
    while ((1LL << label_bits) <= K) ++label_bits;               // labels 0 .. K (K = out of range)
    const int grid = cdiv((long)n, 256);
    for (int c = 0; c < C; ++c) {
        unsigned int *cnt = c == 0 ? counts : nullptr;
        if (c == 0) HIP_TRY(hipMemsetAsync(counts, 0, ((size_t)K + 1) * 4, st));
        if (dtype == DT_U8) hipLaunchKernelGGL(k_median_keys<uint8_t>, grid, 256, 0, st, (const uint8_t *)img, C, c, n, labels, K, keyA, labA, cnt);
        else if (dtype == DT_F32) hipLaunchKernelGGL(k_median_keys<float>, grid, 256, 0, st, (const float *)img, C, c, n, labels, K, keyA, labA, cnt);
        else hipLaunchKernelGGL(k_median_keys<double>, grid, 256, 0, st, (const double *)img, C, c, n, labels, K, keyA, labA, cnt);
        if (c == 0) {
            size_t t = tmp_bytes;
            HIP_TRY(hipcub::DeviceScan::ExclusiveSum(b, t, counts, offsets, K, st));
        }
        size_t t = tmp_bytes;
        HIP_TRY(hipcub::DeviceRadixSort::SortPairs(b, t, keyA, keyB, labA, labB, (int)n, 0, 64, st));
        t = tmp_bytes;
        HIP_TRY(hipcub::DeviceRadixSort::SortPairs(b, t, labB, labA, keyB, keyA, (int)n, 0, label_bits, st));
        hipLaunchKernelGGL(k_median_pick, cdiv(K, 256), 256, 0, st, keyA, offsets, counts, K, C, c, dtype, out);
    }
    HIP_TRY(hipGetLastError());
    return 0;
}

}  // namespace imsegm
