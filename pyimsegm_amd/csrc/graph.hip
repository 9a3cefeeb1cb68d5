// graph.hip -- superpixel adjacency graph, superpixel centres and the final LUT gathers.
//
// Replaces (reference, per-pixel Python / numpy):
//   make_graph_segm_connect_grid2d_conn4 + make_graph_segment_connect_edges
//       /root/reference/imsegm/superpixels.py:115-177   (dict relabel loop, sort, hash, np.unique)
//   superpixel_centers                                   superpixels.py:205-242 (regionprops)
//   proba[slic], graph_labels[slic]                      pipelines.py:104,109
//
// Adjacency: every pixel compares its label with the right and lower neighbour; differing pairs
// set one bit in a K x K bitmap (row = larger id b, column = smaller id a).  Reading the bitmap in
// row-major order yields the edges sorted by (b, a) -- exactly the order the reference obtains from
// np.unique(a + nb_vertices * b).  Centres are exact int64 coordinate sums / counts.
#include "slic.h"

namespace imsegm {

constexpr int GR_ROWS = 4;        // rows per wave (one pixel column per lane); a workgroup covers 64 x 16 pixels
constexpr int GR_SLOTS = 64;      // LDS hash slots for the centre sums of a workgroup

// Edges: a pixel is compared with its right neighbour (the next lane; the pixel right of the wave for lane 63) and the
// one below (the next row of the same lane; one halo row per wave).  Centre sums: the horizontal runs of equal labels of a
// 64-pixel row come out of one ballot; the first lane of a run knows its length, hence n, sum x and sum y of the run in
// closed form, and adds them to the workgroup's LDS hash table (label -> n, sum y, sum x; 32-bit LDS atomics), which is
// flushed with one set of int64 global atomics per label and workgroup.  No wave reductions, no per-label passes.  A
// label that finds no slot (more than GR_SLOTS labels in a 64 x 16 tile) goes to the global sums directly.
__global__ void __launch_bounds__(256)
k_adjacency_centres(const int32_t *__restrict__ labels, int H, int W, int K, int words, uint32_t *bitmap,
                    long long *__restrict__ cacc, size_t zs)
{
    ZSHIFT(labels, zs); ZSHIFT(bitmap, zs); ZSHIFT(cacc, zs);
    __shared__ int h_key[GR_SLOTS], h_n[GR_SLOTS], h_sy[GR_SLOTS], h_sx[GR_SLOTS];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x < GR_SLOTS) {
        h_key[threadIdx.x] = -1;
        h_n[threadIdx.x] = 0;
        h_sy[threadIdx.x] = 0;
        h_sx[threadIdx.x] = 0;
    }
    __syncthreads();
    const int x = blockIdx.x * 64 + lane;
    const int y0 = (blockIdx.y * 4 + wave) * GR_ROWS;
    const bool xin = x < W;
    int lab[GR_ROWS + 1], right[GR_ROWS];
#pragma unroll
    for (int r = 0; r <= GR_ROWS; ++r) lab[r] = (xin && y0 + r < H) ? labels[(size_t)(y0 + r) * W + x] : -1;
#pragma unroll
    for (int r = 0; r < GR_ROWS; ++r) {
        right[r] = __shfl_down(lab[r], 1, 64);
        if (lane == 63) right[r] = (x + 1 < W && y0 + r < H) ? labels[(size_t)(y0 + r) * W + x + 1] : -1;
    }
    const unsigned long long le = (lane == 63) ? ~0ULL : ((2ULL << lane) - 1ULL);
#pragma unroll
    for (int r = 0; r < GR_ROWS; ++r) {
        const int y = y0 + r;
        const int l = lab[r];
        const int left = __shfl_up(l, 1, 64);
        const bool act = l >= 0;
        // (wave-uniform from here on only through the ballots)
        if (act) {
            const int nb[2] = { right[r], lab[r + 1] };       // -1: no neighbour on that side
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                if (nb[j] < 0 || nb[j] == l) continue;
                const int a = min(l, nb[j]), b = max(l, nb[j]);
                uint32_t *wp = bitmap + (size_t)b * words + (a >> 5);
                const uint32_t bit = 1u << (a & 31);
                if (!(*wp & bit)) atomicOr(wp, bit);
            }
        }
        const bool start = act && (lane == 0 || left != l);
        const unsigned long long starts = __ballot(start);
        const int nact = __popcll(__ballot(act));
        if (start) {
            const unsigned long long above = starts & ~le;
            const int len = (above ? __ffsll((long long)above) - 1 : nact) - lane;
            const int n = len, sy = len * y, sx = len * x + (len * (len - 1)) / 2;
            int slot = (int)(((unsigned int)l * 2654435761u) >> 26);          // 6 bits
            bool placed = false;
            for (int probe = 0; probe < GR_SLOTS; ++probe) {
                const int old = atomicCAS(&h_key[slot], -1, l);
                if (old == -1 || old == l) {
                    placed = true;
                    break;
                }
                slot = (slot + 1) & (GR_SLOTS - 1);
            }
            if (placed) {
                atomicAdd(&h_n[slot], n);
                atomicAdd(&h_sy[slot], sy);
                atomicAdd(&h_sx[slot], sx);
            } else {
                atomic_add_i64(cacc + (size_t)l * 3 + 0, n);
                atomic_add_i64(cacc + (size_t)l * 3 + 1, (long long)sy);
                atomic_add_i64(cacc + (size_t)l * 3 + 2, (long long)sx);
            }
        }
    }
    __syncthreads();
    if (threadIdx.x < GR_SLOTS && h_key[threadIdx.x] >= 0) {
        const int k = h_key[threadIdx.x];
        atomic_add_i64(cacc + (size_t)k * 3 + 0, h_n[threadIdx.x]);
        atomic_add_i64(cacc + (size_t)k * 3 + 1, h_sy[threadIdx.x]);
        atomic_add_i64(cacc + (size_t)k * 3 + 2, h_sx[threadIdx.x]);
    }
}

__global__ void k_centres_finalize(const long long *__restrict__ cacc, int K, double *centres, uint8_t *present, size_t zs)
{
    ZSHIFT(cacc, zs); ZSHIFT(centres, zs); ZSHIFT(present, zs);
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= K) return;
    long long n = cacc[(size_t)k * 3];
    present[k] = n > 0;
    // labels without pixels -> [-1, -1] (superpixels.py:218)
    centres[2 * k + 0] = n > 0 ? i64_to_double(cacc[(size_t)k * 3 + 1]) / (double)n : -1.0;
    centres[2 * k + 1] = n > 0 ? i64_to_double(cacc[(size_t)k * 3 + 2]) / (double)n : -1.0;
}

__global__ void k_edge_rowcount(const uint32_t *__restrict__ bitmap, int K, int words, int32_t *rowcount)
{
    int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= K) return;
    int c = 0;
    for (int w = 0; w < words; ++w) c += __popc(bitmap[(size_t)b * words + w]);
    rowcount[b] = c;
}

// exclusive scan of rowcount by one workgroup (K is small); total -> n_edges
__global__ void __launch_bounds__(256) k_edge_scan(int32_t *rowcount, int K, int32_t *n_edges)
{
    __shared__ int wsum[4];
    __shared__ int carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int base = 0; base < K; base += 256) {
        int i = base + threadIdx.x;
        int v = i < K ? rowcount[i] : 0;
        int incl = v;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            int t = __shfl_up(incl, off, 64);
            if (lane >= off) incl += t;
        }
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        int pre = carry;
        for (int w = 0; w < wave; ++w) pre += wsum[w];
        if (i < K) rowcount[i] = pre + incl - v;
        __syncthreads();
        if (threadIdx.x == 255) carry = pre + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) *n_edges = carry;
}

__global__ void k_edge_emit(const uint32_t *__restrict__ bitmap, int K, int words, const int32_t *__restrict__ offsets,
                            int32_t *edges, int capacity)
{
    int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= K) return;
    int o = offsets[b];
    for (int w = 0; w < words; ++w) {
        uint32_t bits = bitmap[(size_t)b * words + w];
        while (bits) {
            int t = __ffs(bits) - 1;
            bits &= bits - 1;
            if (o < capacity) {
                edges[2 * o + 0] = w * 32 + t;
                edges[2 * o + 1] = b;
            }
            ++o;
        }
    }
}

// ---- edges out of the neighbour table (volume.hip k_vol_adjacency_runs<1>): row b holds ALL neighbours of b in any order with free
// slots (-1) in between; one wave per row counts the smaller ones / writes them in ascending order (the rank of an entry is the
// number of smaller ones), which gives the edge order of the bitmap: sorted by (b, a)
__global__ void __launch_bounds__(256) k_table_rowcount(const int32_t *__restrict__ table, int K, int cap, int32_t *rowcount)
{
    const int lane = threadIdx.x & 63;
    const int b = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (b >= K) return;
    int c = 0;
    for (int i = lane; i < cap; i += 64) {
        const int a = table[(size_t)b * cap + i];
        c += a >= 0 && a < b;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) c += __shfl_xor(c, off, 64);
    if (lane == 0) rowcount[b] = c;
}

__global__ void __launch_bounds__(256) k_table_emit(const int32_t *__restrict__ table, int K, int cap, const int32_t *__restrict__ offsets,
                                                    int32_t *edges, int capacity)
{
    const int lane = threadIdx.x & 63;
    const int b = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (b >= K) return;
    const int32_t *row = table + (size_t)b * cap;
    const int o = offsets[b];
    for (int i = lane; i < cap; i += 64) {
        const int a = row[i];
        if (a < 0 || a >= b) continue;                   // (the larger neighbours of b are the edges of THEIR rows)
        int rank = 0;
        for (int j = 0; j < cap; ++j) {
            const int other = row[j];
            rank += other >= 0 && other < a;
        }
        if (o + rank < capacity) {
            edges[2 * (o + rank) + 0] = a;
            edges[2 * (o + rank) + 1] = b;
        }
    }
}

int launch_edge_extract_table(const int32_t *table, int K, int cap, int32_t *rowcount, int32_t *edges_out, int edge_capacity,
                              int32_t *n_edges_dev, hipStream_t st)
{
    hipLaunchKernelGGL(k_table_rowcount, cdiv((long)K * 64, 256), 256, 0, st, table, K, cap, rowcount);
    hipLaunchKernelGGL(k_edge_scan, 1, 256, 0, st, rowcount, K, n_edges_dev);
    hipLaunchKernelGGL(k_table_emit, cdiv((long)K * 64, 256), 256, 0, st, table, K, cap, rowcount, edges_out, edge_capacity);
    HIP_TRY(hipGetLastError());
    return 0;
}

// zero `bytes` bytes (a multiple of 4, 4-byte aligned) of every image of a batch in one launch
__global__ void __launch_bounds__(256) k_zero_words(uint32_t *p, size_t words, size_t zs)
{
    ZSHIFT(p, zs);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < words; i += (size_t)gridDim.x * 256) p[i] = 0u;
}

int launch_zero(void *ptr, size_t bytes, hipStream_t st, ZBatch zb)
{
    if (zb.nz <= 1) {
        HIP_TRY(hipMemsetAsync(ptr, 0, bytes, st));
        return 0;
    }
    const size_t words = (bytes + 3) / 4;
    const int gx = (int)std::min<size_t>(std::max<size_t>(cdiv((long)words, 256 * 8), 1), 256);
    hipLaunchKernelGGL(k_zero_words, dim3(gx, 1, zb.nz), 256, 0, st, static_cast<uint32_t *>(ptr), words, zb.zs);
    HIP_TRY(hipGetLastError());
    return 0;
}

// rows of `words` 32-bit words between a strided and a contiguous layout (the small per-image blocks of a batch: parameters in,
// counters out -- one transfer over the host link for the batch instead of one per image)
__global__ void __launch_bounds__(256) k_copy_rows(uint32_t *dst, size_t dst_stride, const uint32_t *src, size_t src_stride, size_t words)
{
    dst = zshift(dst, dst_stride);
    src = zshift(src, src_stride);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < words; i += (size_t)gridDim.x * 256) dst[i] = src[i];
}

int launch_copy_rows(void *dst, size_t dst_stride, const void *src, size_t src_stride, size_t bytes, int rows, hipStream_t st)
{
    const size_t words = (bytes + 3) / 4;
    const int gx = (int)std::min<size_t>(std::max<size_t>(cdiv((long)words, 256 * 4), 1), 64);
    hipLaunchKernelGGL(k_copy_rows, dim3(gx, 1, rows), 256, 0, st, static_cast<uint32_t *>(dst), dst_stride, static_cast<const uint32_t *>(src),
                       src_stride, words);
    HIP_TRY(hipGetLastError());
    return 0;
}

int launch_adjacency_bitmap(const int32_t *labels, int H, int W, int K, uint32_t *bitmap, long long *cacc, double *centres_out,
                            uint8_t *present_out, hipStream_t st, ZBatch zb)
{
    int words = cdiv(K, 32);
    // (one fill when the sums sit right behind the bitmap, as the fused call lays them out)
    const size_t bm_bytes = (size_t)K * words * sizeof(uint32_t), gap = (size_t)((const char *)cacc - (const char *)bitmap);
    if ((const char *)cacc >= (const char *)bitmap + bm_bytes && gap <= bm_bytes + 64) {
        if (launch_zero(bitmap, gap + (size_t)K * 3 * sizeof(long long), st, zb)) return -1;
    } else {
        if (launch_zero(bitmap, bm_bytes, st, zb) || launch_zero(cacc, (size_t)K * 3 * sizeof(long long), st, zb)) return -1;
    }
    dim3 grid(cdiv(W, 64), cdiv(H, 4 * GR_ROWS), zb.nz);
    hipLaunchKernelGGL(k_adjacency_centres, grid, 256, 0, st, labels, H, W, K, words, bitmap, cacc, zb.zs);
    hipLaunchKernelGGL(k_centres_finalize, dim3(cdiv(K, 256), 1, zb.nz), 256, 0, st, cacc, K, centres_out, present_out, zb.zs);
    HIP_TRY(hipGetLastError());
    return 0;
}

int launch_adjacency_centres(const int32_t *labels, int H, int W, int K, uint32_t *bitmap, long long *cacc,
                             int32_t *edges_out, int edge_capacity, int32_t *n_edges_dev, double *centres_out,
                             uint8_t *present_out, int32_t *rowcount, hipStream_t st)
{
    if (launch_adjacency_bitmap(labels, H, W, K, bitmap, cacc, centres_out, present_out, st, ZBatch())) return -1;
    return launch_edge_extract(bitmap, K, cdiv(K, 32), rowcount, edges_out, edge_capacity, n_edges_dev, st);
}

// read the K x K adjacency bitmap out in row-major order: edges [a, b], a < b, sorted by (b, a)
int launch_edge_extract(const uint32_t *bitmap, int K, int words, int32_t *rowcount, int32_t *edges_out, int edge_capacity,
                        int32_t *n_edges_dev, hipStream_t st)
{
    hipLaunchKernelGGL(k_edge_rowcount, cdiv(K, 256), 256, 0, st, bitmap, K, words, rowcount);
    hipLaunchKernelGGL(k_edge_scan, 1, 256, 0, st, rowcount, K, n_edges_dev);
    hipLaunchKernelGGL(k_edge_emit, cdiv(K, 256), 256, 0, st, bitmap, K, words, rowcount, edges_out, edge_capacity);
    HIP_TRY(hipGetLastError());
    return 0;
}

// ---- LUT gathers --------------------------------------------------------------------------------------
// (four indices per lane where the arrays are 16-byte aligned: one 16-byte load and store instead of four of each -- 2.67 -> see
// profiles/README_r06.md for the 2^30 voxels of config 5)
__global__ void __launch_bounds__(256)
k_gather_i32(const int32_t *__restrict__ lut, const int32_t *__restrict__ idx, size_t n, int32_t *__restrict__ out, size_t zs)
{
    ZSHIFT(lut, zs); ZSHIFT(idx, zs); ZSHIFT(out, zs);
    const size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i >= n) return;
    const bool aligned = ((reinterpret_cast<uintptr_t>(idx) | reinterpret_cast<uintptr_t>(out)) & 15) == 0;
    if (aligned && i + 4 <= n) {
        const int4 q = *reinterpret_cast<const int4 *>(idx + i);
        *reinterpret_cast<int4 *>(out + i) = make_int4(lut[q.x], lut[q.y], lut[q.z], lut[q.w]);
    } else {
        for (size_t j = i; j < n && j < i + 4; ++j) out[j] = lut[idx[j]];
    }
}

__global__ void __launch_bounds__(256)
k_gather_f64(const double *__restrict__ lut, int C, const int32_t *__restrict__ idx, size_t total,
             double *__restrict__ out)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    size_t p = i / C;
    int c = (int)(i - p * C);
    out[i] = lut[(size_t)idx[p] * C + c];
}

int launch_gather_labels(const int32_t *lut, const int32_t *idx, size_t n, int32_t *out, hipStream_t st, ZBatch zb)
{
    hipLaunchKernelGGL(k_gather_i32, dim3(cdiv(cdiv((long)n, 4), 256), 1, zb.nz), 256, 0, st, lut, idx, n, out, zb.zs);
    HIP_TRY(hipGetLastError());
    return 0;
}

int launch_gather_proba(const double *lut, int C, const int32_t *idx, size_t n, double *out, hipStream_t st)
{
    size_t total = n * C;
    hipLaunchKernelGGL(k_gather_f64, cdiv((long)total, 256), 256, 0, st, lut, C, idx, total, out);
    HIP_TRY(hipGetLastError());
    return 0;
}

}  // namespace imsegm
