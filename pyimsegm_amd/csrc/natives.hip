// natives.hip -- the two remaining natives of the reference's Cython module, for BATCHES of positions (one launch):
//   computeLabelHistogram2d      /root/reference/imsegm/features_cython.pyx:222-241   (descriptors.py:1411-1495)
//   computeRayFeaturesBinary2d   features_cython.pyx:244-282                          (descriptors.py:1630-1660)
// The reference calls them once per position from Python loops (compute_label_hist_proba, compute_ray_features_positions,
// the region-growing and centre-detection tools); here one wave serves one position.
#include "slic.h"

namespace imsegm {

// hist[p][l] = number of pixels of window p (segm[y0 : y0 + h, x0 : x0 + w]) with label l >= 0 where the structuring
// element (rows from sy0, columns from sx0) equals 1.  Integer atomics: exact, order independent.
__global__ void __launch_bounds__(256)
k_label_hist2d(const int16_t *__restrict__ segm, int H, int W, const int32_t *__restrict__ windows, int P,
               const int16_t *__restrict__ selem, int SH, int SW, int nb_labels, unsigned int *__restrict__ hist)
{
    const int p = blockIdx.x;
    if (p >= P) return;
    const int y0 = windows[6 * p + 0], x0 = windows[6 * p + 1], h = windows[6 * p + 2], w = windows[6 * p + 3];
    const int sy0 = windows[6 * p + 4], sx0 = windows[6 * p + 5];
    const int n = h * w;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const int r = i / w, c = i - r * w;
        const int y = y0 + r, x = x0 + c, sy = sy0 + r, sx = sx0 + c;
        if (y < 0 || y >= H || x < 0 || x >= W || sy < 0 || sy >= SH || sx < 0 || sx >= SW) continue;
        const int l = segm[(size_t)y * W + x];
        if (l >= 0 && l < nb_labels && selem[(size_t)sy * SW + sx] == 1) atomicAdd(&hist[(size_t)p * nb_labels + l], 1u);
    }
}

// Ray features: from every position a ray per angle walks the binary segmentation in unit steps of the larger
// direction component until it meets the searched edge ('up' = 1: first foreground pixel, 'down' = -1: first
// background pixel after foreground); float32 arithmetic in the operation order of the .pyx (the per-angle direction
// table -- sin / cos of the float32 angle divided by its larger component -- comes from the host, formed with the
// same numpy calls the reference makes).  One wave per position, one lane per angle.
__global__ void __launch_bounds__(64)
k_ray_features_binary2d(const int8_t *__restrict__ seg, int H, int W, const int32_t *__restrict__ positions, int P,
                        const float *__restrict__ grad, int A, int edge, float *__restrict__ out)
{
    const int p = blockIdx.x;
    if (p >= P) return;
    const int py = positions[2 * p], px = positions[2 * p + 1];
    const bool inside = py >= 0 && py < H && px >= 0 && px < W;
    const int8_t start = inside ? seg[(size_t)py * W + px] : 0;
    const int diag = (int)sqrt((double)W * W + (double)H * H);
    for (int a = threadIdx.x; a < A; a += blockDim.x) {
        float res = -1.f;
        if (start && edge == 1) {
            res = 0.f;                                       // the position lies inside the border label
        } else if (inside) {
            const float g0 = grad[2 * a], g1 = grad[2 * a + 1];
            float pos0 = (float)py, pos1 = (float)px;
            int8_t last = start;
            for (int it = 0; it < diag; ++it) {
                pos0 += g0;
                pos1 += g1;
                const double r0 = round((double)pos0), r1 = round((double)pos1);
                if (pos0 < 0 || r0 >= H || pos1 < 0 || r1 >= W) break;
                const int8_t actual = seg[(size_t)(int)r0 * W + (int)r1];
                if ((edge == 1 && actual) || (edge == -1 && last && !actual)) {
                    const float dx = pos0 - (float)py, dy = pos1 - (float)px;
                    res = (float)sqrt((double)((dx * dx) + (dy * dy)));
                    break;
                }
                last = actual;
            }
        }
        out[(size_t)p * A + a] = res;
    }
}

int launch_label_hist2d(const int16_t *segm, int H, int W, const int32_t *windows, int P, const int16_t *selem, int SH, int SW,
                        int nb_labels, unsigned int *hist, hipStream_t st)
{
    HIP_TRY(hipMemsetAsync(hist, 0, (size_t)P * nb_labels * sizeof(unsigned int), st));
    if (P > 0) hipLaunchKernelGGL(k_label_hist2d, P, 256, 0, st, segm, H, W, windows, P, selem, SH, SW, nb_labels, hist);
    HIP_TRY(hipGetLastError());
    return 0;
}

int launch_ray_features_binary2d(const int8_t *seg, int H, int W, const int32_t *positions, int P, const float *grad, int A, int edge,
                                 float *out, hipStream_t st)
{
    if (P > 0) hipLaunchKernelGGL(k_ray_features_binary2d, P, 64, 0, st, seg, H, W, positions, P, grad, A, edge, out);
    HIP_TRY(hipGetLastError());
    return 0;
}

}  // namespace imsegm
