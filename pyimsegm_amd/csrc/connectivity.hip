// connectivity.hip -- `_enforce_label_connectivity_cython` (skimage/segmentation/_slic.pyx 0.18,
// reached through /root/reference/imsegm/superpixels.py:61-63 `enforce_connectivity=True`)
// re-designed for the GPU, bit-identical to the sequential raster-scan algorithm
// (oracle: orc_enforce_connectivity).
//
// The sequential algorithm visits pixels in raster order; every still-unlabelled pixel seeds a BFS
// over its 4-connected equal-label component, capped at max_size pixels; a component smaller than
// min_size is merged into `adjacent` (label of the last already-labelled foreign neighbour met by
// the BFS, 0 if none), otherwise it receives the next consecutive label.
//
// Parallel formulation used here:
//   1. union-find CCL with min-raster-index roots  ->  the root IS the seed pixel, and components
//      are processed by the sequential algorithm in increasing root order;
//   2. a component >= max_size is truncated to the first max_size pixels of its BFS order (exact
//      sequential BFS by one thread, rare), the left-over pixels are re-labelled (CCL again) and
//      become later components of their own -- repeated until no oversize component is left;
//   3. kept components (size >= min_size) get label = start + rank among kept roots (prefix sum);
//   4. a small component needs the exact BFS discovery order only to find WHICH already-labelled
//      neighbour is met last: "already labelled" == belongs to a component with a smaller root,
//      which is static, so all small components run their (tiny) BFS concurrently, one thread each;
//   5. merged-into-merged chains are resolved by pointer chasing (roots strictly decrease).
#include "slic.h"

namespace imsegm {

enum { ST_ACTIVE = 0, ST_FINAL = 1 };
enum { CNT_OVER = 0, CNT_SMALL = 1, CNT_CURSOR = 2, CNT_KEPT = 3, CNT_OVERLIST = 4 };

__device__ __forceinline__ int uf_find(const int32_t *parent, int a)
{
    int p = parent[a];
    while (p != a) {
        a = p;
        p = parent[a];
    }
    return a;
}

__device__ __forceinline__ void uf_union(int32_t *parent, int a, int b)
{
    while (true) {
        a = uf_find(parent, a);
        b = uf_find(parent, b);
        if (a == b) return;
        if (a < b) {
            int t = a;
            a = b;
            b = t;
        }
        // a > b: hang the larger root below the smaller one
        int old = atomicMin(&parent[a], b);
        if (old == a) return;
        a = old;
    }
}

__global__ void __launch_bounds__(256) k_ccl_init(int32_t *parent, const uint8_t *state, int n)
{
    int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p < n && state[p] == ST_ACTIVE) parent[p] = p;
}

__global__ void __launch_bounds__(256)
k_ccl_merge(const int32_t *__restrict__ labels, const uint8_t *__restrict__ state, int32_t *parent, int H, int W)
{
    int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= H * W || state[p] != ST_ACTIVE) return;
    int y = p / W, x = p - y * W;
    int l = labels[p];
    if (x > 0 && state[p - 1] == ST_ACTIVE && labels[p - 1] == l) uf_union(parent, p, p - 1);
    if (y > 0 && state[p - W] == ST_ACTIVE && labels[p - W] == l) uf_union(parent, p, p - W);
}

__global__ void __launch_bounds__(256) k_ccl_flatten(int32_t *parent, const uint8_t *state, int32_t *csize, int n)
{
    int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n || state[p] != ST_ACTIVE) return;
    int r = uf_find(parent, p);
    parent[p] = r;
    if (r == p) csize[p] = 0;
}

// component sizes: wave-aggregated atomics (runs of equal roots are the common case)
__global__ void __launch_bounds__(256)
k_comp_size(const int32_t *__restrict__ parent, const uint8_t *__restrict__ state, int32_t *csize, int n)
{
    int p = blockIdx.x * blockDim.x + threadIdx.x;
    int r = (p < n && state[p] == ST_ACTIVE) ? parent[p] : -1;
    while (true) {
        unsigned long long vote = __ballot(r >= 0);
        if (!vote) break;
        int leader = __ffsll((long long)vote) - 1;
        int lr = __shfl(r, leader, 64);
        unsigned long long same = __ballot(r == lr);
        if ((threadIdx.x & 63) == leader) atomicAdd(&csize[lr], __popcll(same));
        if (r == lr) r = -1;
    }
}

__global__ void __launch_bounds__(256)
k_find_oversize(const int32_t *__restrict__ parent, const uint8_t *__restrict__ state, const int32_t *__restrict__ csize,
                int n, int max_size, int32_t *over_list, int32_t *counters)
{
    int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n || state[p] != ST_ACTIVE || parent[p] != p) return;
    if (csize[p] >= max_size) {
        int i = atomicAdd(&counters[CNT_OVER], 1);
        over_list[i] = p;
    }
}

// exact sequential BFS of the reference, capped at max_size discovered pixels (one thread per oversize root)
__global__ void k_oversize_bfs(const int32_t *over_list, int n_over, const int32_t *__restrict__ parent,
                               const uint8_t *__restrict__ state, int H, int W, int max_size, int32_t *queue,
                               uint8_t *visited, int32_t *counters)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_over) return;
    int root = over_list[i];
    int base = atomicAdd(&counters[CNT_CURSOR], max_size);
    int32_t *q = queue + base;
    q[0] = root;
    visited[root] = 1;
    int qs = 1;
    for (int v = 0; v < qs && qs < max_size; ++v) {
        int p = q[v];
        int y = p / W, x = p - y * W;
        int nb[4] = { x + 1 < W ? p + 1 : -1, x > 0 ? p - 1 : -1, y + 1 < H ? p + W : -1, y > 0 ? p - W : -1 };
        for (int j = 0; j < 4; ++j) {
            int t = nb[j];
            if (t < 0) continue;
            if (state[t] == ST_ACTIVE && parent[t] == root && !visited[t]) {
                visited[t] = 1;
                q[qs++] = t;
                if (qs >= max_size) break;
            }
        }
    }
}

// after the truncation: chosen pixels + all pixels of regular components become final; left-over
// pixels of oversize components stay active for the next CCL round
__global__ void __launch_bounds__(256)
k_oversize_commit(int32_t *parent, uint8_t *state, int32_t *csize_next, const int32_t *__restrict__ csize,
                  const uint8_t *__restrict__ visited, int n, int max_size)
{
    int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n || state[p] != ST_ACTIVE) return;
    int r = parent[p];
    if (csize[r] >= max_size) {
        if (visited[p]) {
            state[p] = ST_FINAL;
            if (p == r) csize_next[p] = max_size;
        }   // else: stays active, re-initialised by the next k_ccl_init
    } else {
        state[p] = ST_FINAL;
        if (p == r) csize_next[p] = csize[p];
    }
}

// ---- consecutive labels for kept components: rank of kept roots in raster order -----------------
constexpr int SCAN_PER_THREAD = 16;
constexpr int SCAN_BLOCK = 256 * SCAN_PER_THREAD;

__device__ __forceinline__ int block_exclusive_scan(int v, int *total)
{
    __shared__ int wsum[4];
    int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int incl = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        int t = __shfl_up(incl, off, 64);
        if (lane >= off) incl += t;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    int base = 0;
    for (int w = 0; w < wave; ++w) base += wsum[w];
    *total = wsum[0] + wsum[1] + wsum[2] + wsum[3];
    __syncthreads();
    return base + incl - v;
}

template <bool ASSIGN>
__global__ void __launch_bounds__(256)
k_kept_scan(const int32_t *__restrict__ parent, const int32_t *__restrict__ csize, int n, int min_size,
            int32_t *blocksum, int32_t *newlabel, int start_label)
{
    int p0 = blockIdx.x * SCAN_BLOCK + threadIdx.x * SCAN_PER_THREAD;
    int cnt = 0;
    for (int j = 0; j < SCAN_PER_THREAD; ++j) {
        int p = p0 + j;
        if (p < n && parent[p] == p && csize[p] >= min_size) cnt++;
    }
    int total;
    int excl = block_exclusive_scan(cnt, &total);
    if (!ASSIGN) {
        if (threadIdx.x == 0) blocksum[blockIdx.x] = total;
    } else {
        int rank = blocksum[blockIdx.x] + excl;
        for (int j = 0; j < SCAN_PER_THREAD; ++j) {
            int p = p0 + j;
            if (p < n && parent[p] == p && csize[p] >= min_size) newlabel[p] = start_label + rank++;
        }
    }
}

// exclusive scan of the per-block counts by one workgroup; total -> counters[CNT_KEPT]
__global__ void __launch_bounds__(256) k_scan_blocksums(int32_t *blocksum, int nblocks, int32_t *counters)
{
    __shared__ int carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int base = 0; base < nblocks; base += 256) {
        int i = base + threadIdx.x;
        int v = i < nblocks ? blocksum[i] : 0;
        int total;
        int excl = block_exclusive_scan(v, &total);
        if (i < nblocks) blocksum[i] = carry + excl;
        __syncthreads();
        if (threadIdx.x == 0) carry += total;
        __syncthreads();
    }
    if (threadIdx.x == 0) counters[CNT_KEPT] = carry;
}

// ---- small components ------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_list_small(const int32_t *__restrict__ parent, const int32_t *__restrict__ csize, int n, int min_size,
             int32_t *list, int32_t *counters)
{
    int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n || parent[p] != p) return;
    if (csize[p] < min_size) {
        int i = atomicAdd(&counters[CNT_SMALL], 1);
        list[i] = p;
    }
}

// one thread = one small component: the reference's BFS (neighbour order +x, -x, +y, -y), recording
// the component of the last already-labelled (smaller root) foreign neighbour
__global__ void __launch_bounds__(64)
k_small_bfs(const int32_t *__restrict__ list, const int32_t *__restrict__ counters, const int32_t *__restrict__ parent,
            const int32_t *__restrict__ csize, int H, int W, int32_t *queue, uint8_t *visited, int32_t *cursor,
            int32_t *adjptr)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= counters[CNT_SMALL]) return;
    int root = list[i];
    int size = csize[root];
    int adj = -1;
    if (size == 1) {
        int p = root;
        int y = p / W, x = p - y * W;
        int nb[4] = { x + 1 < W ? p + 1 : -1, x > 0 ? p - 1 : -1, y + 1 < H ? p + W : -1, y > 0 ? p - W : -1 };
        for (int j = 0; j < 4; ++j)
            if (nb[j] >= 0) {
                int c = parent[nb[j]];
                if (c < root) adj = c;
            }
        adjptr[root] = adj;
        return;
    }
    int base = atomicAdd(cursor, size);
    int32_t *q = queue + base;
    q[0] = root;
    visited[root] = 1;
    int qs = 1;
    for (int v = 0; v < qs; ++v) {
        int p = q[v];
        int y = p / W, x = p - y * W;
        int nb[4] = { x + 1 < W ? p + 1 : -1, x > 0 ? p - 1 : -1, y + 1 < H ? p + W : -1, y > 0 ? p - W : -1 };
        for (int j = 0; j < 4; ++j) {
            int t = nb[j];
            if (t < 0) continue;
            int c = parent[t];
            if (c == root) {
                if (!visited[t]) {
                    visited[t] = 1;
                    q[qs++] = t;
                }
            } else if (c < root) {
                adj = c;
            }
        }
    }
    adjptr[root] = adj;
}

__global__ void __launch_bounds__(64)
k_small_resolve(const int32_t *__restrict__ list, const int32_t *__restrict__ counters,
                const int32_t *__restrict__ csize, const int32_t *__restrict__ adjptr, int min_size,
                int32_t *newlabel)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= counters[CNT_SMALL]) return;
    int root = list[i];
    int r = adjptr[root];
    while (r >= 0 && csize[r] < min_size) r = adjptr[r];
    // `adjacent = 0` when the BFS met no labelled neighbour (_slic.pyx); r is a kept root here
    newlabel[root] = (r >= 0) ? newlabel[r] : 0;
}

__global__ void __launch_bounds__(256)
k_write_labels(const int32_t *__restrict__ parent, const int32_t *__restrict__ newlabel, int n, int32_t *out)
{
    int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    out[p] = newlabel[parent[p]];
}

int launch_enforce_connectivity(const int32_t *labels_in, int H, int W, long min_size_l, long max_size_l,
                                int start_label, ConnWork w, int32_t *labels_out, int *n_labels_out_host,
                                hipStream_t st)
{
    const int n = H * W;
    const int grid = cdiv(n, 256);
    const int min_size = (int)std::min<long>(min_size_l, 0x7fffffff);
    const int max_size = (int)std::min<long>(max_size_l, 0x7fffffff);
    uint8_t *state = w.visited + n;       // second half of the byte scratch (2 * n bytes)
    int32_t *csize_final = w.adjptr;      // reused: final sizes are gathered here during the rounds
    HIP_TRY(hipMemsetAsync(state, ST_ACTIVE, n, st));
    HIP_TRY(hipMemsetAsync(w.counters, 0, 16 * sizeof(int32_t), st));
    int32_t host_counters[16];
    for (int round = 0;; ++round) {
        HIP_TRY(hipMemsetAsync(w.visited, 0, n, st));
        HIP_TRY(hipMemsetAsync(w.counters, 0, 3 * sizeof(int32_t), st));
        hipLaunchKernelGGL(k_ccl_init, grid, 256, 0, st, w.parent, state, n);
        hipLaunchKernelGGL(k_ccl_merge, grid, 256, 0, st, labels_in, state, w.parent, H, W);
        hipLaunchKernelGGL(k_ccl_flatten, grid, 256, 0, st, w.parent, state, w.csize, n);
        hipLaunchKernelGGL(k_comp_size, grid, 256, 0, st, w.parent, state, w.csize, n);
        hipLaunchKernelGGL(k_find_oversize, grid, 256, 0, st, w.parent, state, w.csize, n, max_size, w.list, w.counters);
        HIP_TRY(hipMemcpyAsync(host_counters, w.counters, 4 * sizeof(int32_t), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        int n_over = host_counters[CNT_OVER];
        if (n_over > 0) {
            if ((long)n_over * max_size > (long)n) {
                set_error("enforce_connectivity: internal queue overflow");
                return -1;
            }
            hipLaunchKernelGGL(k_oversize_bfs, cdiv(n_over, 64), 64, 0, st, w.list, n_over, w.parent, state, H, W,
                               max_size, w.queue, w.visited, w.counters);
        }
        hipLaunchKernelGGL(k_oversize_commit, grid, 256, 0, st, w.parent, state, csize_final, w.csize, w.visited, n,
                           max_size);
        if (n_over == 0) break;
        if (round > n) {
            set_error("enforce_connectivity: did not converge");
            return -1;
        }
    }
    // from here on: parent[p] = root of the final component, csize_final[root] = its size
    const int nblocks = cdiv(n, SCAN_BLOCK);
    hipLaunchKernelGGL(k_kept_scan<false>, nblocks, 256, 0, st, w.parent, csize_final, n, min_size, w.blocksum,
                       w.newlabel, start_label);
    hipLaunchKernelGGL(k_scan_blocksums, 1, 256, 0, st, w.blocksum, nblocks, w.counters);
    hipLaunchKernelGGL(k_kept_scan<true>, nblocks, 256, 0, st, w.parent, csize_final, n, min_size, w.blocksum,
                       w.newlabel, start_label);
    HIP_TRY(hipMemsetAsync(w.visited, 0, n, st));
    HIP_TRY(hipMemsetAsync(w.counters + CNT_SMALL, 0, 2 * sizeof(int32_t), st));
    hipLaunchKernelGGL(k_list_small, grid, 256, 0, st, w.parent, csize_final, n, min_size, w.list, w.counters);
    // the number of small components is only known on the device: launch for the worst case the
    // grid can hold cheaply and let surplus threads exit (list length is read from counters)
    HIP_TRY(hipMemcpyAsync(host_counters, w.counters, 4 * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    int n_small = host_counters[CNT_SMALL];
    int n_kept = host_counters[CNT_KEPT];
    if (n_small > 0) {
        // adjptr aliases csize_final, so the BFS writes its result into w.csize (free now)
        hipLaunchKernelGGL(k_small_bfs, cdiv(n_small, 64), 64, 0, st, w.list, w.counters, w.parent, csize_final, H, W,
                           w.queue, w.visited, w.counters + CNT_CURSOR, w.csize);
        hipLaunchKernelGGL(k_small_resolve, cdiv(n_small, 64), 64, 0, st, w.list, w.counters, csize_final, w.csize,
                           min_size, w.newlabel);
    }
    hipLaunchKernelGGL(k_write_labels, grid, 256, 0, st, w.parent, w.newlabel, n, labels_out);
    HIP_TRY(hipGetLastError());
    *n_labels_out_host = n_kept > 0 ? start_label + n_kept : 1;
    return 0;
}

}  // namespace imsegm
