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

#include <atomic>

namespace imsegm {

enum { ST_ACTIVE = 0, ST_FINAL = 1 };
enum { CNT_OVER = 0, CNT_SMALL = 1, CNT_CURSOR = 2, CNT_KEPT = 3, CNT_FALLBACK = 4, CNT_BIG = 5, CNT_LITTLE = 6,
       CNT_LROOT = 8, CNT_FLAG = 9, CNT_FB = 10, CNT_FB2 = 11 };      // 8..11: the 2-D tile path

// neighbour of voxel p in direction d of the reference's BFS order (+x, -x, +y, -y, +z, -z); -1 outside
__device__ __forceinline__ int neighbour(int p, int D, int H, int W, int d)
{
    const int x = p % W, y = (p / W) % H, z = p / (W * H);
    switch (d) {
        case 0: return x + 1 < W ? p + 1 : -1;
        case 1: return x > 0 ? p - 1 : -1;
        case 2: return y + 1 < H ? p + W : -1;
        case 3: return y > 0 ? p - W : -1;
        case 4: return z + 1 < D ? p + W * H : -1;
        default: return z > 0 ? p - W * H : -1;
    }
}

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

// union with the two finds walked together (both loads of a step in flight at once: half the dependent trips of uf_union)
__device__ __forceinline__ void uf_union_pair(int32_t *parent, int a, int b)
{
    while (true) {
        while (true) {
            const int pa = parent[a], pb = parent[b];
            if (pa == a && pb == b) break;
            a = pa;
            b = pb;
        }
        if (a == b) return;
        if (a < b) {
            const int t = a;
            a = b;
            b = t;
        }
        const int old = atomicMin(&parent[a], b);
        if (old == a) return;
        a = old;
    }
}

// Run-based initialisation: every active pixel points at the leftmost pixel of its horizontal run
// inside its 64-pixel wave segment (ballot + count-leading-zeros, no memory traffic), so the
// union-find forest starts with paths of length <= 1 and the merge pass only has to join runs.
// ALL: every pixel is active (the first round, i.e. always on the fast path): the state bytes are not read
template <bool ALL> __device__ __forceinline__ bool is_active(const uint8_t *state, int p)
{
    return ALL || state[p] == ST_ACTIVE;
}

template <bool ALL>
__global__ void __launch_bounds__(256)
k_ccl_init(const int32_t *__restrict__ labels, const uint8_t *__restrict__ state, int32_t *parent, int n, int W)
{
    int p = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    bool active = p < n && is_active<ALL>(state, p);
    bool cont = false;
    if (active) cont = (p % W) > 0 && is_active<ALL>(state, p - 1) && labels[p - 1] == labels[p];
    unsigned long long starts = __ballot(active && !cont);
    if (!active) return;
    unsigned long long below = starts & ((lane == 63) ? ~0ULL : ((2ULL << lane) - 1ULL));
    int start_lane = below ? 63 - __clzll((long long)below) : 0;
    parent[p] = p - (lane - start_lane);
}

template <bool ALL>
__global__ void __launch_bounds__(256)
k_ccl_merge(const int32_t *__restrict__ labels, const uint8_t *__restrict__ state, int32_t *parent, int D, int H, int W)
{
    int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= D * H * W || !is_active<ALL>(state, p)) return;
    int x = p % W, y = (p / W) % H, z = p / (W * H);
    int l = labels[p];
    bool cont = x > 0 && is_active<ALL>(state, p - 1) && labels[p - 1] == l;
    // horizontal: only the first lane of a wave segment still has to be tied to its left neighbour
    if (cont && (threadIdx.x & 63) == 0) uf_union(parent, p, p - 1);
    // vertical: one union per pair of overlapping runs is enough -- skip it when the left
    // neighbour pair (p-1, p-W-1) carries the same two runs (it, or a pixel further left, does it)
    if (y > 0 && is_active<ALL>(state, p - W) && labels[p - W] == l) {
        bool left_pair_same = cont && is_active<ALL>(state, p - W - 1) && labels[p - W - 1] == l;
        if (!left_pair_same) uf_union(parent, p, p - W);
    }
    const int HW = H * W;
    if (z > 0 && is_active<ALL>(state, p - HW) && labels[p - HW] == l) {
        bool left_pair_same = cont && is_active<ALL>(state, p - HW - 1) && labels[p - HW - 1] == l;
        if (!left_pair_same) uf_union(parent, p, p - HW);
    }
}

// Round 6, first round of a volume (every voxel active): initialisation and merge pass by row segments.  A wave covers CR_SPAN = 62
// voxels of a row with lane 0 / lane 63 carrying the voxels left / right of them; the three rows the rule looks at -- (z, y),
// (z, y - 1), (z - 1, y) -- are loaded once per wave, the x - 1 neighbours come from the neighbouring lane (one DPP move), and the
// coordinates from the grid (no division).  Same rule as k_ccl_init / k_ccl_merge above (runs inside a segment start as one set; one
// union per pair of overlapping runs), hence the same components with the same roots (the smallest index of a set).
constexpr int CR_SPAN = 62;

// (four rows per wave here, as in the merge: 3.77 against 3.27 ms at config 5 -- not kept)
__global__ void __launch_bounds__(256)
k_ccl_init_rows(const int32_t *__restrict__ labels, int32_t *__restrict__ parent, int32_t *__restrict__ csize, int H, int W)
{
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int x = (blockIdx.x * 4 + wave) * CR_SPAN + lane - 1;
    const size_t row = ((size_t)blockIdx.z * H + blockIdx.y) * W;
    const bool inx = x >= 0 && x < W;
    const int l = inx ? labels[row + x] : -5;            // (a voxel no window covered carries -1: the fillers differ from it)
    const bool mine = inx && lane >= 1 && lane <= CR_SPAN;
    const bool cont = lane_prev(l, -8) == l && lane > 1;
    const unsigned long long starts = __ballot(mine && !cont);
    if (!mine) return;
    const unsigned long long below = starts & ((2ULL << lane) - 1ULL);     // (lane <= 62)
    const int start_lane = 63 - __clzll((long long)below);
    parent[row + x] = (int)(row + x) - (lane - start_lane);
    if (start_lane == lane) csize[row + x] = 0;          // (a root is the first voxel of a run: the sizes k_ccl_flatten_sizes adds up)
}

// CR_ROWS rows of the slice per wave: the row above the first and the rows behind are loaded once for all of them, and the unions
// of a lane over its rows -- a bit each in `todo`: 3 r + 0 left (a run that crosses into the segment), + 1 up, + 2 behind -- are
// done two at a time (union2_min_root), every lane that still has some side by side.  (One row per wave, one union per lane and
// round: 10.4 ms at 2^30 voxels, the latency of one chain of dependent loads after the other.)
#ifndef CCL_MERGE_ROWS
#define CCL_MERGE_ROWS 4
#endif
constexpr int CR_ROWS = CCL_MERGE_ROWS;          // (<= 10: three bits of `todo` per row)

__global__ void __launch_bounds__(256)
k_ccl_merge_rows(const int32_t *__restrict__ labels, int32_t *parent, int D, int H, int W)
{
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // (an eighth of every slice's row groups per XCD -- linear block index modulo 8 -> a fixed range of rows, so that a component's
    // forest is walked out of ONE L2 -- was measured: 9.2 against 8.7 ms at config 5, one box; the blocks stay in launch order)
    const int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    const int x = (bx * 4 + wave) * CR_SPAN + lane - 1;
    const int y0 = by * CR_ROWS, z = bz;
    const bool inx = x >= 0 && x < W;
    const int plane = H * W;
    const size_t row0 = (size_t)z * plane + (size_t)y0 * W;
    int l[CR_ROWS + 1], bk[CR_ROWS];                          // l[0]: the row above the first; fillers differ from every label (>= -1)
    l[0] = (inx && y0 > 0) ? labels[row0 - W + x] : -6;
#pragma unroll
    for (int r = 0; r < CR_ROWS; ++r) {
        const bool rowok = y0 + r < H;
        l[1 + r] = (inx && rowok) ? labels[row0 + (size_t)r * W + x] : -5;
        bk[r] = (inx && rowok && z > 0) ? labels[row0 - plane + (size_t)r * W + x] : -7;
    }
    int lp[CR_ROWS + 1];
#pragma unroll
    for (int r = 0; r <= CR_ROWS; ++r) lp[r] = lane_prev(l[r], -8);
    const bool seg = inx && lane >= 1 && lane <= CR_SPAN;
    unsigned todo = 0;
#pragma unroll
    for (int r = 0; r < CR_ROWS; ++r) {
        const int cur = l[1 + r];
        const bool mine = seg && y0 + r < H;
        const bool cont = lp[1 + r] == cur;                    // the left neighbour carries the same label
        const bool up_left = lp[r] == cur, back_left = lane_prev(bk[r], -8) == cur;
        // one union per pair of overlapping runs: skipped when the pair to the left (p - 1, q - 1) carries the same two runs
        if (mine && cont && lane == 1) todo |= 1u << (3 * r);
        if (mine && l[r] == cur && !(cont && up_left)) todo |= 2u << (3 * r);
        if (mine && bk[r] == cur && !(cont && back_left)) todo |= 4u << (3 * r);
    }
    // (starting a union from the first voxels of the two runs -- found by votes over the rows in registers instead of one load
    // each -- was measured: 9.4 against 8.7 ms; the first step of a walk hits the cache, the votes cost more)
    const int p0 = (int)(row0 + x);
    while (__any(todo != 0)) {
        int a[2] = { -1, -1 }, b[2] = { -1, -1 };
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            if (!todo) continue;
            const int bit = __ffs(todo) - 1;
            todo &= todo - 1;
            const int r = bit / 3, k = bit - 3 * r;
            a[j] = p0 + r * W;
            b[j] = a[j] - (k == 0 ? 1 : k == 1 ? W : plane);
        }
        union2_min_root(parent, a[0], b[0], a[1], b[1]);
    }
}

template <bool ALL>
__global__ void __launch_bounds__(256) k_ccl_flatten(int32_t *parent, const uint8_t *state, int32_t *csize, int n)
{
    int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n || !is_active<ALL>(state, p)) return;
    int r = uf_find(parent, p);
    parent[p] = r;
    if (r == p) csize[p] = 0;
}

// Round 6, first round of a volume: flatten and component sizes in one pass (csize zeroed at the run starts by k_ccl_init_rows).
// Four voxels per lane (16-byte load / store), their walks to the root side by side; the sizes by RUNS of equal roots along the
// 256 voxels of a wave -- a lane whose four voxels agree joins the run of its left neighbour, the first lane of a run adds the
// run's length with one atomic; a lane with a root change inside adds its pieces itself.
__global__ void __launch_bounds__(256) k_ccl_flatten_sizes(int32_t *parent, int32_t *csize, int n)
{
    const int lane = threadIdx.x & 63;
    const int p = (blockIdx.x * 256 + threadIdx.x) * 4;
    const bool in = p < n, full = p + 4 <= n;
    int r[4];
    if (full) {
        const int4 q = *reinterpret_cast<const int4 *>(parent + p);
        r[0] = q.x; r[1] = q.y; r[2] = q.z; r[3] = q.w;
    } else {
#pragma unroll
        for (int c = 0; c < 4; ++c) r[c] = p + c < n ? parent[p + c] : 0;
    }
    while (true) {
        int q[4];
        bool moved = false;
#pragma unroll
        for (int c = 0; c < 4; ++c) q[c] = parent[r[c]];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            moved |= q[c] != r[c];
            r[c] = q[c];
        }
        if (!moved) break;
    }
    if (full) {
        *reinterpret_cast<int4 *>(parent + p) = make_int4(r[0], r[1], r[2], r[3]);
    } else {
        for (int c = 0; p + c < n; ++c) parent[p + c] = r[c];
    }
    const bool uniform = full && r[0] == r[1] && r[1] == r[2] && r[2] == r[3];
    const int key = uniform ? r[0] : -1 - lane;                       // (a mixed lane is a run of its own)
    const bool start = lane == 0 || lane_prev(key, -100) != key;
    const unsigned long long starts = __ballot(start);
    if (uniform && start) {
        const unsigned long long above = lane == 63 ? 0ULL : starts >> (lane + 1);
        const int lanes = above ? __ffsll((long long)above) : 64 - lane;
        atomicAdd(&csize[r[0]], 4 * lanes);
    }
    if (in && !uniform) {
        int run = 0;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            if (p + c >= n) break;
            run++;
            if (c == 3 || p + c + 1 >= n || r[c + 1] != r[c]) {
                atomicAdd(&csize[r[c]], run);
                run = 0;
            }
        }
    }
}

// The same pass by tiles of 256 x 16 voxels of a slice (W a multiple of four: the rows stay 16-byte aligned): a wave takes four rows,
// the sixteen walks of a lane side by side, and the runs add their lengths to the tile's roots in an LDS table first -- a tile meets a
// dozen components, its rows a hundred runs --; one global atomic per root and tile at the end.  (Every run its own global atomic:
// 70 * 10^6 of them at config 5, half of the 7.3 ms of the pass.)
constexpr int FS_SLOTS = 64;

__device__ __forceinline__ void fs_add(int *keys, int *vals, int32_t *csize, int root, int count)
{
    int slot = (int)(((unsigned int)root * 2654435761u) >> 26);          // 6 bits
    for (int probe = 0; probe < FS_SLOTS; ++probe) {
        const int old = atomicCAS(&keys[slot], -1, root);
        if (old == -1 || old == root) {
            atomicAdd(&vals[slot], count);
            return;
        }
        slot = (slot + 1) & (FS_SLOTS - 1);
    }
    atomicAdd(&csize[root], count);                                        // (more components than slots in one tile)
}

__global__ void __launch_bounds__(256) k_ccl_flatten_sizes_rows(int32_t *parent, int32_t *csize, int H, int W)
{
    __shared__ int keys[FS_SLOTS], vals[FS_SLOTS];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (threadIdx.x < FS_SLOTS) {
        keys[threadIdx.x] = -1;
        vals[threadIdx.x] = 0;
    }
    __syncthreads();
    const int x = (blockIdx.x * 64 + lane) * 4;
    const int y0 = blockIdx.y * 16 + wave * 4;
    const size_t slice = (size_t)blockIdx.z * H * W;
    const bool inx = x < W;                                                // (W % 4 == 0: the four voxels of a lane are in or out together)
    int r[4][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const bool ok = inx && y0 + j < H;
        const int4 q = ok ? *reinterpret_cast<const int4 *>(parent + slice + (size_t)(y0 + j) * W + x) : make_int4(0, 0, 0, 0);
        r[j][0] = q.x; r[j][1] = q.y; r[j][2] = q.z; r[j][3] = q.w;
    }
    while (true) {
        bool moved = false;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int q[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) q[c] = parent[r[j][c]];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                moved |= q[c] != r[j][c];
                r[j][c] = q[c];
            }
        }
        if (!moved) break;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const bool ok = inx && y0 + j < H;                                   // (wave uniform in y)
        if (ok) *reinterpret_cast<int4 *>(parent + slice + (size_t)(y0 + j) * W + x) = make_int4(r[j][0], r[j][1], r[j][2], r[j][3]);
        const bool uniform = ok && r[j][0] == r[j][1] && r[j][1] == r[j][2] && r[j][2] == r[j][3];
        const int key = uniform ? r[j][0] : -1 - lane;                      // (a mixed lane is a run of its own)
        const bool start = lane == 0 || lane_prev(key, -100) != key;
        const unsigned long long starts = __ballot(start);
        if (uniform && start) {
            const unsigned long long above = lane == 63 ? 0ULL : starts >> (lane + 1);
            const int lanes = above ? __ffsll((long long)above) : 64 - lane;
            fs_add(keys, vals, csize, r[j][0], 4 * lanes);
        }
        if (ok && !uniform) {
            int run = 0;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                run++;
                if (c == 3 || r[j][c + 1] != r[j][c]) {
                    fs_add(keys, vals, csize, r[j][c], run);
                    run = 0;
                }
            }
        }
    }
    __syncthreads();
    if (threadIdx.x < FS_SLOTS && keys[threadIdx.x] >= 0) atomicAdd(&csize[keys[threadIdx.x]], vals[threadIdx.x]);
}

// component sizes: wave-aggregated atomics (runs of equal roots are the common case)
template <bool ALL>
__global__ void __launch_bounds__(256)
k_comp_size(const int32_t *__restrict__ parent, const uint8_t *__restrict__ state, int32_t *csize, int n)
{
    int p = blockIdx.x * blockDim.x + threadIdx.x;
    int r = (p < n && is_active<ALL>(state, p)) ? parent[p] : -1;
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

template <bool ALL>
__global__ void __launch_bounds__(256)
k_find_oversize(const int32_t *__restrict__ parent, const uint8_t *__restrict__ state, const int32_t *__restrict__ csize,
                int n, int max_size, int32_t *over_list, int32_t *counters)
{
    int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n || !is_active<ALL>(state, p) || parent[p] != p) return;
    if (csize[p] >= max_size) {
        int i = atomicAdd(&counters[CNT_OVER], 1);
        over_list[i] = p;
    }
}

// exact sequential BFS of the reference, capped at max_size discovered pixels (one thread per oversize root)
__global__ void k_oversize_bfs(const int32_t *over_list, int n_over, const int32_t *__restrict__ parent,
                               const uint8_t *__restrict__ state, int D, int H, int W, int max_size, int32_t *queue,
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
        for (int j = 0; j < 6; ++j) {
            int t = neighbour(p, D, H, W, j);
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
// A workgroup takes SCAN_BLOCK voxels as SCAN_TILES tiles of 1 024: four consecutive voxels per lane through one 16-byte load, a
// wave reads 1 KB contiguous (until round 6 a lane took 16 consecutive voxels one by one -- every load instruction of a wave touched
// 64 cache lines, and the assigning pass read them twice: 2.3 + 4.2 ms at 2^30 voxels).
constexpr int SCAN_TILES = 4;
constexpr int SCAN_BLOCK = SCAN_TILES * 1024;

template <int NW> __device__ __forceinline__ int block_exclusive_scan(int v, int *total)
{
    __shared__ int wsum[NW];
    int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int incl = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        int t = __shfl_up(incl, off, 64);
        if (lane >= off) incl += t;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    int base = 0, all = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
        base += w < wave ? wsum[w] : 0;
        all += wsum[w];
    }
    *total = all;
    __syncthreads();
    return base + incl - v;
}

// !ASSIGN: first pass -- kept roots counted per workgroup; components that reach `max_size` raise counters[CNT_OVER] (the fast
//          path's check that no component has to be truncated -- k_find_oversize in the rounds of the general path);
// ASSIGN:  second pass -- kept roots receive their labels, the roots of small components are appended to `list` (one atomic per
//          wave that has any; the order of the list is free -- k_small_order sorts it by size class)
template <bool ASSIGN>
__global__ void __launch_bounds__(256)
k_kept_scan(const int32_t *__restrict__ parent, const int32_t *__restrict__ csize, int n, int min_size, int max_size,
            int32_t *blocksum, int32_t *newlabel, int start_label, int32_t *list, int32_t *counters)
{
    const int base = blockIdx.x * SCAN_BLOCK + threadIdx.x * 4;
    unsigned kept = 0, small = 0;                   // bit 4 * tile + c
    int cnt[SCAN_TILES], n_small = 0, n_over = 0;
#pragma unroll
    for (int i = 0; i < SCAN_TILES; ++i) {
        const int p = base + i * 1024;
        int v[4];
        if (p + 4 <= n) {
            const int4 q = *reinterpret_cast<const int4 *>(parent + p);
            v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
        } else {
#pragma unroll
            for (int c = 0; c < 4; ++c) v[c] = p + c < n ? parent[p + c] : -1;
        }
        cnt[i] = 0;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            if (v[c] != p + c) continue;
            const int size = csize[p + c];
            if (size >= min_size) {
                kept |= 1u << (4 * i + c);
                cnt[i]++;
            } else {
                small |= 1u << (4 * i + c);
                n_small++;
            }
            n_over += size >= max_size;
        }
    }
    if (!ASSIGN) {
        int total;
        block_exclusive_scan<4>(cnt[0] + cnt[1] + cnt[2] + cnt[3], &total);
        if (threadIdx.x == 0) blocksum[blockIdx.x] = total;
        if (n_over) atomicAdd(&counters[CNT_OVER], n_over);
    } else {
        // the small roots of the wave behind one another in the list
        int incl = n_small;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int t = __shfl_up(incl, off, 64);
            if ((int)(threadIdx.x & 63) >= off) incl += t;
        }
        const int wave_total = __shfl(incl, 63, 64);
        int at = 0;
        if (wave_total) {
            if ((threadIdx.x & 63) == 63) at = atomicAdd(&counters[CNT_SMALL], wave_total);
            at = __shfl(at, 63, 64) + incl - n_small;
        }
        int rank0 = blocksum[blockIdx.x];
#pragma unroll
        for (int i = 0; i < SCAN_TILES; ++i) {
            int total;
            int rank = rank0 + block_exclusive_scan<4>(cnt[i], &total);
            rank0 += total;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int p = base + i * 1024 + c;
                if (kept >> (4 * i + c) & 1u) newlabel[p] = start_label + rank++;
                if (small >> (4 * i + c) & 1u) list[at++] = p;
            }
        }
    }
}

// exclusive scan of the per-block counts by one workgroup; total -> counters[CNT_KEPT]
__global__ void __launch_bounds__(1024) k_scan_blocksums(int32_t *blocksum, int nblocks, int32_t *counters)
{
    // (four consecutive counts per lane and turn: a turn costs a trip to memory and its barriers whatever it carries)
    constexpr int PER = 4;
    __shared__ int carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int base = 0; base < nblocks; base += 1024 * PER) {
        const int i = base + threadIdx.x * PER;
        int v[PER], t = 0;
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            v[j] = i + j < nblocks ? blocksum[i + j] : 0;
            t += v[j];
        }
        int total;
        int excl = carry + block_exclusive_scan<16>(t, &total);
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            if (i + j < nblocks) blocksum[i + j] = excl;
            excl += v[j];
        }
        __syncthreads();
        if (threadIdx.x == 0) carry += total;
        __syncthreads();
    }
    if (threadIdx.x == 0) counters[CNT_KEPT] = carry;
}

// ---- small components ------------------------------------------------------------------------------
__global__ void __launch_bounds__(64)
k_small_resolve(const int32_t *__restrict__ list, const int32_t *__restrict__ counters,
                const int32_t *__restrict__ csize, const int32_t *__restrict__ adjptr, int min_size,
                int32_t *newlabel)
{
    const int n_small = counters[CNT_FLAG] ? 0 : counters[CNT_SMALL];
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n_small; i += gridDim.x * blockDim.x) {
        int root = list[i];
        int r = adjptr[root];
        while (r >= 0 && csize[r] < min_size) r = adjptr[r];
        // `adjacent = 0` when the BFS met no labelled neighbour (_slic.pyx); r is a kept root here
        newlabel[root] = (r >= 0) ? newlabel[r] : 0;
    }
}

// ---- cooperative BFS: one wave per small component -----------------------------------------------
// bounding boxes of the small components (only their few pixels take part)
__global__ void __launch_bounds__(256)
k_small_bbox_init(int32_t *bbox, const int32_t *__restrict__ list, const int32_t *__restrict__ counters,
                  int32_t *slotmap, int capacity)
{
    const int n_small = min(counters[CNT_SMALL], capacity);
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n_small; i += gridDim.x * blockDim.x) {
        bbox[6 * i + 0] = 0x7fffffff;   // min y
        bbox[6 * i + 1] = -1;           // max y
        bbox[6 * i + 2] = 0x7fffffff;   // min x
        bbox[6 * i + 3] = -1;           // max x
        bbox[6 * i + 4] = 0x7fffffff;   // min z
        bbox[6 * i + 5] = -1;           // max z
        slotmap[list[i]] = i;
    }
}

// Bounding boxes of the small components.  Four consecutive pixels per lane (one 16-byte load of their roots): the
// roots nearly always agree, so a lane looks up one component size and builds one box per run of equal small roots;
// the boxes of a wave that belong to the same component are merged with shuffles before one lane issues the atomics
// (all pixels of a small component used to hammer the same six words).
struct Box6 {
    int y0, y1, x0, x1, z0, z1;
};

__device__ __forceinline__ void box_commit_wave(int slot, Box6 b, int32_t *bbox)
{
    // lanes with slot < 0 have nothing; leader loop over the distinct slots of the wave
    while (true) {
        const unsigned long long vote = __ballot(slot >= 0);
        if (!vote) break;
        const int leader = __ffsll((long long)vote) - 1;
        const int ls = __shfl(slot, leader, 64);
        const bool mine = slot == ls;
        Box6 m;
        m.y0 = mine ? b.y0 : 0x7fffffff; m.y1 = mine ? b.y1 : -1;
        m.x0 = mine ? b.x0 : 0x7fffffff; m.x1 = mine ? b.x1 : -1;
        m.z0 = mine ? b.z0 : 0x7fffffff; m.z1 = mine ? b.z1 : -1;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            m.y0 = min(m.y0, __shfl_xor(m.y0, off, 64)); m.y1 = max(m.y1, __shfl_xor(m.y1, off, 64));
            m.x0 = min(m.x0, __shfl_xor(m.x0, off, 64)); m.x1 = max(m.x1, __shfl_xor(m.x1, off, 64));
            m.z0 = min(m.z0, __shfl_xor(m.z0, off, 64)); m.z1 = max(m.z1, __shfl_xor(m.z1, off, 64));
        }
        if ((int)(threadIdx.x & 63) == leader) {
            atomicMin(&bbox[6 * ls + 0], m.y0);
            atomicMax(&bbox[6 * ls + 1], m.y1);
            atomicMin(&bbox[6 * ls + 2], m.x0);
            atomicMax(&bbox[6 * ls + 3], m.x1);
            atomicMin(&bbox[6 * ls + 4], m.z0);
            atomicMax(&bbox[6 * ls + 5], m.z1);
        }
        if (mine) slot = -1;
    }
}

__global__ void __launch_bounds__(256)
k_small_bbox(const int32_t *__restrict__ parent, const int32_t *__restrict__ csize, int n, int H, int W, int min_size,
             const int32_t *__restrict__ slotmap, int32_t *bbox, const int32_t *__restrict__ counters, int capacity)
{
    if (counters[CNT_SMALL] > capacity) return;         // table too small: the thread-BFS fallback takes all (uniform)
    const int p0 = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
    int r[4] = { -1, -1, -1, -1 };
    if (p0 + 4 <= n) {
        const int4 v = *reinterpret_cast<const int4 *>(parent + p0);
        r[0] = v.x; r[1] = v.y; r[2] = v.z; r[3] = v.w;
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) r[j] = p0 + j < n ? parent[p0 + j] : -1;
    }
    // runs of equal small roots inside the lane: at most four boxes, nearly always one or none
    int slot[4];
    Box6 box[4];
    int last_r = -1, last_slot = -1;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        slot[j] = -1;
        box[j] = { 0x7fffffff, -1, 0x7fffffff, -1, 0x7fffffff, -1 };
        if (r[j] < 0) continue;
        if (r[j] != last_r) {
            last_r = r[j];
            last_slot = csize[last_r] < min_size ? slotmap[last_r] : -1;
        }
        if (last_slot < 0) continue;
        slot[j] = last_slot;
        const int p = p0 + j;
        const int x = p % W, y = (p / W) % H, z = p / (W * H);
        box[j] = { y, y, x, x, z, z };
    }
    // fold every run into its first pixel (static indices only: the boxes stay in registers)
#pragma unroll
    for (int j = 2; j >= 0; --j)
        if (slot[j] >= 0 && slot[j + 1] == slot[j]) {
            box[j].y0 = min(box[j].y0, box[j + 1].y0); box[j].y1 = max(box[j].y1, box[j + 1].y1);
            box[j].x0 = min(box[j].x0, box[j + 1].x0); box[j].x1 = max(box[j].x1, box[j + 1].x1);
            box[j].z0 = min(box[j].z0, box[j + 1].z0); box[j].z1 = max(box[j].z1, box[j + 1].z1);
        }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int sl = (j == 0 || slot[j] != slot[j - 1]) ? slot[j] : -1;      // run start
        if (!__any(sl >= 0)) continue;
        box_commit_wave(sl, box[j], bbox);
    }
}

// Processing order of the small components for k_small_bfs_wave: the ones with a large bounding box (long
// BFS, the critical path of the launch) first, the many tiny ones after them.
__global__ void __launch_bounds__(256)
k_small_order(const int32_t *__restrict__ bbox, int32_t *counters, int D, int H, int W, int big_cells, int capacity,
              int32_t *order)
{
    const int n_small = counters[CNT_SMALL];
    if (n_small > capacity) return;
    for (int ci = blockIdx.x * blockDim.x + threadIdx.x; ci < n_small; ci += gridDim.x * blockDim.x) {
        const int y0 = max(bbox[6 * ci + 0] - 1, 0), y1 = min(bbox[6 * ci + 1] + 1, H - 1);
        const int x0 = max(bbox[6 * ci + 2] - 1, 0), x1 = min(bbox[6 * ci + 3] + 1, W - 1);
        const int z0 = max(bbox[6 * ci + 4] - 1, 0), z1 = min(bbox[6 * ci + 5] + 1, D - 1);
        const long cells = (long)(x1 - x0 + 1) * (y1 - y0 + 1) * (z1 - z0 + 1);
        if (cells > big_cells) order[atomicAdd(&counters[CNT_BIG], 1)] = ci;
        else order[n_small - 1 - atomicAdd(&counters[CNT_LITTLE], 1)] = ci;
    }
}

// Exact emulation of the reference's BFS (queue order, neighbour order +x, -x, +y, -y) by one wave:
// level-synchronous, the frontier is kept in BFS-rank order.  Frontier element i proposes key
// 4*i + d to each unvisited member neighbour (LDS atomicMin); the minimum key of a cell is its
// first discovery, and compacting the winning keys in ascending order (ballot + popcount) yields
// the next frontier in exactly the order the sequential queue would hold it.  `adjacent` is the
// component of the foreign, already-labelled (smaller root) neighbour with the largest
// (level, 4*i + d).  The bounding box (+1 ring) of the component is staged in LDS first.
// HOP2 (2-D tile path below): parent[] holds the tile-local root of a pixel, whose own entry is the global root.
// `order` may be null (identity); `count` points at the number of listed components.
template <int CELLS, int FMAX, bool HOP2>
__global__ void __launch_bounds__(64)
k_small_bfs_wave(const int32_t *__restrict__ list, int32_t *counters, const int32_t *__restrict__ count,
                 const int32_t *__restrict__ parent, const int32_t *__restrict__ bbox, int D, int H, int W,
                 const int32_t *__restrict__ order, int capacity, int32_t *adjptr, int32_t *fallback_list, size_t zs)
{
    ZSHIFT(list, zs); ZSHIFT(counters, zs); ZSHIFT(count, zs); ZSHIFT(parent, zs); ZSHIFT(bbox, zs); ZSHIFT(order, zs); ZSHIFT(adjptr, zs);
    ZSHIFT(fallback_list, zs);
    __shared__ unsigned int prop[CELLS];
    __shared__ uint8_t cls[CELLS];       // 0 other, 1 member unvisited, 2 earlier foreign, 3 member visited
    __shared__ uint16_t fr[2][FMAX];
    __shared__ int level_best;
    static_assert(CELLS <= 8192 && 6 * FMAX <= (1 << 18), "packing of (key, cell) in level_best");
    const int lane = threadIdx.x;
    const int n_small = *count;
    if (n_small > capacity || (HOP2 && counters[CNT_FLAG])) return;
    for (int t = blockIdx.x; t < n_small; t += gridDim.x) {
        const int ci = order ? order[t] : t;
        const int root = list[ci];
        const int y0 = max(bbox[6 * ci + 0] - 1, 0), y1 = min(bbox[6 * ci + 1] + 1, H - 1);
        const int x0 = max(bbox[6 * ci + 2] - 1, 0), x1 = min(bbox[6 * ci + 3] + 1, W - 1);
        const int z0 = max(bbox[6 * ci + 4] - 1, 0), z1 = min(bbox[6 * ci + 5] + 1, D - 1);
        const int bw = x1 - x0 + 1, bh = y1 - y0 + 1, bd = z1 - z0 + 1;
        const int bwh = bw * bh;
        const long cells_l = (long)bwh * bd;
        if (cells_l > CELLS) {
            if (lane == 0) fallback_list[atomicAdd(&counters[CNT_FALLBACK], 1)] = root;
            continue;
        }
        const int cells = (int)cells_l;
        __syncthreads();
        for (int c = lane; c < cells; c += 64) {
            int cz = c / bwh, rem = c - cz * bwh;
            int cy = rem / bw, cx = rem - cy * bw;
            int q = parent[((size_t)(z0 + cz) * H + (y0 + cy)) * W + x0 + cx];
            if (HOP2) q = parent[q];
            cls[c] = (q == root) ? 1 : (q < root ? 2 : 0);
            prop[c] = 0xffffffffu;
        }
        const int rx = root % W, ry = (root / W) % H, rz = root / (W * H);
        const int seed = ((rz - z0) * bh + (ry - y0)) * bw + (rx - x0);
        // exact division of a cell index (< 2^13) by bw / bwh through a 32-bit reciprocal
        const unsigned int m_bw = (unsigned int)((0x100000000ull + bw - 1) / bw);
        const unsigned int m_bwh = (unsigned int)((0x100000000ull + bwh - 1) / bwh);
        __syncthreads();
        if (lane == 0) {
            fr[0][0] = (uint16_t)seed;
            cls[seed] = 3;
            level_best = -1;
        }
        int f = 1, cur = 0;
        int adj_cell = -1;
        bool failed = false;
        __syncthreads();
        while (f > 0) {
            const int nkeys = 6 * f;
            // phase A: proposals + foreign neighbours; the last foreign contact of this level (largest key)
            // overrides earlier levels: one LDS atomicMax on (key << 13 | cell)
            for (int key = lane; key < nkeys; key += 64) {
                const int c = fr[cur][key / 6], d = key % 6;
                const int cz = bd == 1 ? 0 : (int)__umulhi((unsigned int)c, m_bwh), rem = c - cz * bwh;
                const int cy = bw == 1 ? rem : (int)__umulhi((unsigned int)rem, m_bw), cx = rem - cy * bw;
                const int nx = cx + (d == 0) - (d == 1), ny = cy + (d == 2) - (d == 3), nz = cz + (d == 4) - (d == 5);
                if (nx < 0 || nx >= bw || ny < 0 || ny >= bh || nz < 0 || nz >= bd) continue;
                const int nc = (nz * bh + ny) * bw + nx;
                const int t = cls[nc];
                if (t == 1) atomicMin(&prop[nc], (unsigned int)key);
                else if (t == 2) atomicMax(&level_best, (key << 13) | nc);
            }
            __syncthreads();
            if (level_best >= 0) adj_cell = level_best & 8191;
            // phase B: winners, compacted in key order
            int base = 0;
            for (int k0 = 0; k0 < nkeys; k0 += 64) {
                int key = k0 + lane;
                bool win = false;
                int nc = 0;
                if (key < nkeys) {
                    const int c = fr[cur][key / 6], d = key % 6;
                    const int cz = bd == 1 ? 0 : (int)__umulhi((unsigned int)c, m_bwh), rem = c - cz * bwh;
                    const int cy = bw == 1 ? rem : (int)__umulhi((unsigned int)rem, m_bw), cx = rem - cy * bw;
                    const int nx = cx + (d == 0) - (d == 1), ny = cy + (d == 2) - (d == 3), nz = cz + (d == 4) - (d == 5);
                    if (nx >= 0 && nx < bw && ny >= 0 && ny < bh && nz >= 0 && nz < bd) {
                        nc = (nz * bh + ny) * bw + nx;
                        win = (cls[nc] == 1) && (prop[nc] == (unsigned int)key);
                    }
                }
                unsigned long long m = __ballot(win);
                if (win) {
                    int pos = base + __popcll(m & ((1ULL << lane) - 1ULL));
                    if (pos < FMAX) fr[cur ^ 1][pos] = (uint16_t)nc;
                }
                base += __popcll(m);
            }
            __syncthreads();
            if (base > FMAX) { failed = true; break; }
            for (int i = lane; i < base; i += 64) cls[fr[cur ^ 1][i]] = 3;
            if (lane == 0) level_best = -1;
            __syncthreads();
            f = base;
            cur ^= 1;
        }
        if (failed) {
            if (lane == 0) fallback_list[atomicAdd(&counters[CNT_FALLBACK], 1)] = root;
            continue;
        }
        if (lane == 0) {
            int adj = -1;
            if (adj_cell >= 0) {
                int cz = adj_cell / bwh, rem = adj_cell - cz * bwh;
                int cy = rem / bw, cx = rem - cy * bw;
                adj = parent[((size_t)(z0 + cz) * H + (y0 + cy)) * W + x0 + cx];
                if (HOP2) adj = parent[adj];
            }
            adjptr[root] = adj;
        }
    }
}

// thread-sequential BFS for the components the wave kernel could not take (huge bounding box)
__global__ void __launch_bounds__(64)
k_small_bfs_fallback(const int32_t *__restrict__ list_all, const int32_t *__restrict__ fallback_list,
                     const int32_t *__restrict__ counters, const int32_t *__restrict__ parent,
                     const int32_t *__restrict__ csize, int D, int H, int W, int capacity, int32_t *queue, uint8_t *visited,
                     int32_t *cursor, int32_t *adjptr)
{
    const bool all = counters[CNT_SMALL] > capacity;
    const int count = all ? counters[CNT_SMALL] : counters[CNT_FALLBACK];
    const int32_t *list = all ? list_all : fallback_list;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < count; i += gridDim.x * blockDim.x) {
        int root = list[i];
        int size = csize[root];
        int adj = -1;
        int base = atomicAdd(cursor, size);
        int32_t *q = queue + base;
        q[0] = root;
        visited[root] = 1;
        int qs = 1;
        for (int v = 0; v < qs; ++v) {
            int p = q[v];
            for (int j = 0; j < 6; ++j) {
                int t = neighbour(p, D, H, W, j);
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
}

// (four voxels per lane where the arrays are 16-byte aligned)
__global__ void __launch_bounds__(256)
k_write_labels(const int32_t *__restrict__ parent, const int32_t *__restrict__ newlabel, int n, int32_t *out,
               const int32_t *__restrict__ abort_flag, size_t zs)
{
    ZSHIFT(parent, zs); ZSHIFT(newlabel, zs); ZSHIFT(out, zs); ZSHIFT(abort_flag, zs);
    const int p = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (p >= n || (abort_flag && *abort_flag)) return;
    const bool aligned = ((reinterpret_cast<uintptr_t>(parent) | reinterpret_cast<uintptr_t>(out)) & 15) == 0;
    if (aligned && p + 4 <= n) {
        const int4 q = *reinterpret_cast<const int4 *>(parent + p);
        *reinterpret_cast<int4 *>(out + p) = make_int4(newlabel[q.x], newlabel[q.y], newlabel[q.z], newlabel[q.w]);
    } else {
        for (int j = p; j < n && j < p + 4; ++j) out[j] = newlabel[parent[j]];
    }
}

// ---- skimage.measure.label of the map this file wrote (superpixels.py:104-111: slic, then measure.label) -------------------------
// After the connectivity pass every label > 0 is ONE 6-connected set: a kept component, plus the small components that were merged
// into it through a 6-neighbour (label 0 besides -- start_label 0's first segment and the small components that met no labelled
// neighbour -- is measure.label's background whatever its shape).  A 6-connected set is one component of the full neighbourhood, so
// measure.label has nothing to join: it numbers the labels 1, 2, ... in the raster order of their FIRST voxels and writes 0 for
// label 0.  That is all this does -- first voxel per label, the rank of the first voxels (a bit per voxel, counted per 4 096 and
// scanned), one pass that rewrites the map -- instead of a union-find over 2^30 voxels with thirteen neighbours each (volume.hip
// launch_label_cc, the path of every other label map; tests/test_gpu_volume.py holds the two against each other).
__global__ void __launch_bounds__(256) k_first_fill(int32_t *first, int n_labels)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n_labels) first[i] = 0x7fffffff;
}

// four voxels per lane; only where a run of equal labels starts (in flat order: a later voxel of a run cannot be the first) the
// label's entry is looked at, and only an earlier voxel than the one it holds goes to the atomic -- workgroups start in raster
// order, so next to none do
__global__ void __launch_bounds__(256) k_first_voxel(const int32_t *__restrict__ labels, int n, int n_labels, int32_t *first)
{
    const int p = (blockIdx.x * 256 + threadIdx.x) * 4;
    int v[4];
    if (p + 4 <= n) {
        const int4 q = *reinterpret_cast<const int4 *>(labels + p);
        v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
    } else {
#pragma unroll
        for (int c = 0; c < 4; ++c) v[c] = p + c < n ? labels[p + c] : -1;
    }
    // (the entries of the lane's run starts are requested together, then compared: one trip to memory, not one per voxel)
    int before = lane_prev(v[3], -2);
    int have[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int l = v[c];
        have[c] = -1;                                              // (no run start here: nothing to compare with)
        if (l > 0 && l < n_labels && l != before) have[c] = __hip_atomic_load(first + l, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        before = l;
    }
#pragma unroll
    for (int c = 0; c < 4; ++c)
        if (p + c < have[c]) atomicMin(first + v[c], p + c);
}

__global__ void __launch_bounds__(256) k_first_mark(const int32_t *__restrict__ first, int n_labels, uint32_t *bitmap)
{
    const int l = blockIdx.x * 256 + threadIdx.x;
    if (l < 1 || l >= n_labels) return;
    const int f = first[l];
    if (f != 0x7fffffff) atomicOr(bitmap + (f >> 5), 1u << (f & 31));
}

// set bits per 4 096 voxels (128 words): one wave per block of the scan
__global__ void __launch_bounds__(256) k_first_block_counts(const uint32_t *__restrict__ bitmap, int nblocks, int32_t *blocksum)
{
    const int blk = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (blk >= nblocks) return;
    const uint2 w = *reinterpret_cast<const uint2 *>(bitmap + (size_t)blk * 128 + 2 * lane);
    const int c = wave_sum_i32(__popc(w.x) + __popc(w.y));
    if (lane == 0) blocksum[blk] = c;
}

// first[l] -> the new number of label l (in place): 1 + the first voxels before its own
__global__ void __launch_bounds__(256)
k_first_rank(int32_t *first, int n_labels, const uint32_t *__restrict__ bitmap, const int32_t *__restrict__ blocksum)
{
    const int l = blockIdx.x * 256 + threadIdx.x;
    if (l >= n_labels) return;
    const int f = first[l];
    if (l == 0 || f == 0x7fffffff) {
        first[l] = 0;
        return;
    }
    const int blk = f >> 12, word = f >> 5;
    int rank = blocksum[blk];
    for (int w = blk * 128; w < word; ++w) rank += __popc(bitmap[w]);
    rank += __popc(bitmap[word] & ((1u << (f & 31)) - 1u));
    first[l] = 1 + rank;
}

__global__ void __launch_bounds__(256) k_relabel(int32_t *labels, int n, int n_labels, const int32_t *__restrict__ remap)
{
    const int p = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (p >= n) return;
    if (p + 4 <= n) {
        int4 q = *reinterpret_cast<const int4 *>(labels + p);
        q.x = (q.x > 0 && q.x < n_labels) ? remap[q.x] : 0;
        q.y = (q.y > 0 && q.y < n_labels) ? remap[q.y] : 0;
        q.z = (q.z > 0 && q.z < n_labels) ? remap[q.z] : 0;
        q.w = (q.w > 0 && q.w < n_labels) ? remap[q.w] : 0;
        *reinterpret_cast<int4 *>(labels + p) = q;
    } else {
        for (int c = 0; p + c < n; ++c) {
            const int l = labels[p + c];
            labels[p + c] = (l > 0 && l < n_labels) ? remap[l] : 0;
        }
    }
}

int launch_label_connected(int32_t *labels_inout, size_t n_voxels, int n_labels, int32_t *first, uint32_t *bitmap, int32_t *blocksum,
                           int32_t *counters, int32_t *total_dev, hipStream_t st)
{
    const int n = (int)n_voxels, nblocks = cdiv(n, 4096), quads = cdiv(cdiv(n, 4), 256);
    HIP_TRY(hipMemsetAsync(bitmap, 0, (size_t)nblocks * 512, st));
    hipLaunchKernelGGL(k_first_fill, cdiv(n_labels, 256), 256, 0, st, first, n_labels);
    hipLaunchKernelGGL(k_first_voxel, quads, 256, 0, st, labels_inout, n, n_labels, first);
    hipLaunchKernelGGL(k_first_mark, cdiv(n_labels, 256), 256, 0, st, first, n_labels, bitmap);
    hipLaunchKernelGGL(k_first_block_counts, cdiv(nblocks, 4), 256, 0, st, bitmap, nblocks, blocksum);
    hipLaunchKernelGGL(k_scan_blocksums, 1, 1024, 0, st, blocksum, nblocks, counters);
    hipLaunchKernelGGL(k_first_rank, cdiv(n_labels, 256), 256, 0, st, first, n_labels, bitmap, blocksum);
    hipLaunchKernelGGL(k_relabel, quads, 256, 0, st, labels_inout, n, n_labels, first);
    HIP_TRY(hipMemcpyAsync(total_dev, counters + CNT_KEPT, sizeof(int32_t), hipMemcpyDeviceToDevice, st));
    HIP_TRY(hipGetLastError());
    return 0;
}

// CCL + sizes + oversize detection for the currently active pixels
template <bool ALL>
static void conn_ccl_round(const int32_t *labels_in, int D, int H, int W, int max_size, const ConnWork &w, uint8_t *state,
                           hipStream_t st)
{
    const int n = D * H * W, grid = cdiv(n, 256);
    if (ALL && H <= 65535 && D <= 65535 && !knobs().cc_merge_full) {
        const dim3 rows(cdiv(W, 4 * CR_SPAN), H, D), row_groups(cdiv(W, 4 * CR_SPAN), cdiv(H, CR_ROWS), D);
        hipLaunchKernelGGL(k_ccl_init_rows, rows, 256, 0, st, labels_in, w.parent, w.csize, H, W);
        hipLaunchKernelGGL(k_ccl_merge_rows, row_groups, 256, 0, st, labels_in, w.parent, D, H, W);
        if (W % 4 == 0) hipLaunchKernelGGL(k_ccl_flatten_sizes_rows, dim3(cdiv(W, 256), cdiv(H, 16), D), 256, 0, st, w.parent, w.csize, H, W);
        else hipLaunchKernelGGL(k_ccl_flatten_sizes, cdiv(cdiv(n, 4), 256), 256, 0, st, w.parent, w.csize, n);
        return;
    } else {
        hipLaunchKernelGGL(k_ccl_init<ALL>, grid, 256, 0, st, labels_in, state, w.parent, n, W);
        hipLaunchKernelGGL(k_ccl_merge<ALL>, grid, 256, 0, st, labels_in, state, w.parent, D, H, W);
    }
    hipLaunchKernelGGL(k_ccl_flatten<ALL>, grid, 256, 0, st, w.parent, state, w.csize, n);
    hipLaunchKernelGGL(k_comp_size<ALL>, grid, 256, 0, st, w.parent, state, w.csize, n);
    // (every voxel active = the speculative first round: the first pass of k_kept_scan counts the oversize components)
    if (!ALL) hipLaunchKernelGGL(k_find_oversize<ALL>, grid, 256, 0, st, w.parent, state, w.csize, n, max_size, w.list, w.counters);
}

// everything after the component structure is final: consecutive labels for kept components,
// `adjacent` of the small ones (exact BFS emulation), pointer resolution, label write.
// No host round trip: list lengths stay on the device, the kernels loop over them.
static int conn_tail(const int32_t *csize_final, int32_t *adjptr, int D, int H, int W, int min_size, int start_label,
                     const ConnWork &w, int32_t *labels_out, hipStream_t st, bool counters_are_zero, int oversize_from = 0x7fffffff)
{
    const int n = D * H * W;
    const int nblocks = cdiv(n, SCAN_BLOCK);
    const int capacity = n / 12;                      // bbox table: 6 ints per small component
    int32_t *bbox = w.bbox;
    int32_t *fallback_list = w.bbox + (size_t)6 * capacity;
    if (!counters_are_zero) {
        HIP_TRY(hipMemsetAsync(w.counters + CNT_SMALL, 0, 2 * sizeof(int32_t), st));
        HIP_TRY(hipMemsetAsync(w.counters + CNT_FALLBACK, 0, 3 * sizeof(int32_t), st));     // FALLBACK, BIG, LITTLE
    }
    hipLaunchKernelGGL(k_kept_scan<false>, nblocks, 256, 0, st, w.parent, csize_final, n, min_size, oversize_from, w.blocksum,
                       w.newlabel, start_label, w.list, w.counters);
    hipLaunchKernelGGL(k_scan_blocksums, 1, 1024, 0, st, w.blocksum, nblocks, w.counters);
    hipLaunchKernelGGL(k_kept_scan<true>, nblocks, 256, 0, st, w.parent, csize_final, n, min_size, 0x7fffffff, w.blocksum,
                       w.newlabel, start_label, w.list, w.counters);
    HIP_TRY(hipMemsetAsync(w.visited, 0, n, st));
    hipLaunchKernelGGL(k_small_bbox_init, 64, 256, 0, st, bbox, w.list, w.counters, w.slotmap, capacity);
    hipLaunchKernelGGL(k_small_bbox, cdiv(cdiv(n, 4), 256), 256, 0, st, w.parent, csize_final, n, H, W, min_size, w.slotmap, bbox,
                       w.counters, capacity);
    // one launch for all of them, long ones first (w.queue is free until the fallback kernel)
    hipLaunchKernelGGL(k_small_order, 64, 256, 0, st, bbox, w.counters, D, H, W, 1024, capacity, w.queue);
    hipLaunchKernelGGL((k_small_bfs_wave<8192, 2048, false>), 2048, 64, 0, st, w.list, w.counters, w.counters + CNT_SMALL, w.parent,
                       bbox, D, H, W, w.queue, capacity, adjptr, fallback_list, (size_t)0);
    hipLaunchKernelGGL(k_small_bfs_fallback, 64, 64, 0, st, w.list, fallback_list, w.counters, w.parent, csize_final, D, H,
                       W, capacity, w.queue, w.visited, w.counters + CNT_CURSOR, adjptr);
    hipLaunchKernelGGL(k_small_resolve, 64, 64, 0, st, w.list, w.counters, csize_final, adjptr, min_size, w.newlabel);
    hipLaunchKernelGGL(k_write_labels, cdiv(cdiv(n, 4), 256), 256, 0, st, w.parent, w.newlabel, n, labels_out, (const int32_t *)nullptr, (size_t)0);
    HIP_TRY(hipGetLastError());
    return 0;
}


// ====================================================================================================
// 2-D fast path (every pixel active, no component reaches max_size -- checked on the device).
//
// The label map is read once and the final map written once; everything in between works on the few thousand
// tile-local components instead of the 4 M pixels:
//   k_ccl_tile      one workgroup per 64 x 32 tile: run-based union-find in LDS; parent[p] = tile-local root (global
//                   pixel index of the raster-first pixel of the local component); per local root: size and bounding
//                   box (atomics of the run leaders on <= 64 LDS slots), appended to a dense list
//   k_ccl_border    unions across tile borders (one per pair of touching runs), on the local roots only
//   k_lroot_merge   per local root: global root (min raster index), size and bounding box summed / merged there
//   k_root_classify per global root: kept (>= min_size) -> list of kept roots, small -> list of small roots,
//                   >= max_size -> flag (the general path takes over)
//   k_kept_rank     one workgroup: raster rank of the kept roots (bucket sort in LDS) = their consecutive labels
//   k_small_bfs_reg `adjacent` of every small component: exact emulation of the reference's BFS order by one wave
//                   with the frontier in registers (lane = BFS rank inside the level), the component's bounding box
//                   (+2 rings) staged in LDS as one byte per cell
//   k_small_resolve, k_lroot_labels, k_write_labels   chains of merged components, labels of the local roots, and
//                   out[p] = newlabel[parent[p]]
// Anything the fast path cannot take (more than 64 local components in a tile, list capacities, a frontier of more
// than 64 cells and a bounding box the old wave kernel cannot stage either, an oversize component) raises a flag /
// counter that the host reads at its single synchronisation, and the general path above runs instead.
constexpr int CT_W = 64, CT_H = 32, CT_SLOTS = 256;      // (config-4 images: up to 83 local components in a tile)
constexpr int CONN_KEPT_CAP = 1 << 16, CONN_FB_CAP = 4096;
static_assert(2 * CONN_KEPT_CAP + 16 * CONN_FB_CAP <= CONN_DENSE_INTS, "dense scratch of the tile path");
static_assert(CT_W * CT_H == CONN_TILE_PIXELS && CT_SLOTS == CONN_TILE_SLOTS, "sizes used by conn_i32_bytes (slic.h)");

struct ConnDense {
    int32_t *lroots, *lsize, *lbox, *ntile, *kept, *sorted, *fb_list, *fb_reject, *fb_bbox, *fb2_list, *fb2_bbox;
};
// local roots live in fixed slots: tile t owns entries [t * CT_SLOTS, t * CT_SLOTS + ntile[t]) (no global cursor: two
// thousand workgroups adding to one counter serialise on it)
static ConnDense conn_dense(const ConnWork &w, int n_tiles)
{
    ConnDense d;
    int32_t *b = w.dense;
    d.kept = b; b += CONN_KEPT_CAP;
    d.sorted = b; b += CONN_KEPT_CAP;
    d.fb_list = b; b += CONN_FB_CAP;
    d.fb_reject = b; b += CONN_FB_CAP;
    d.fb_bbox = b; b += 6 * CONN_FB_CAP;
    d.fb2_list = b; b += CONN_FB_CAP;
    d.fb2_bbox = b; b += 6 * CONN_FB_CAP;
    d.lroots = b; b += (size_t)n_tiles * CT_SLOTS;
    d.lsize = b; b += (size_t)n_tiles * CT_SLOTS;
    d.lbox = b; b += (size_t)n_tiles * CT_SLOTS;
    d.ntile = b;
    return d;
}

__device__ __forceinline__ int lds_find(volatile int *par, int a)
{
    int p = par[a];
    while (p != a) {
        a = p;
        p = par[a];
    }
    return a;
}

// (the finds compress: a stack of full-width runs would otherwise grow a chain as long as the tile is high;
// atomicMin keeps the entry a valid, only ever smaller, ancestor)
__device__ __forceinline__ void lds_union(int *par, int a, int b)
{
    while (true) {
        const int a0 = a, b0 = b;
        a = lds_find(par, a);
        b = lds_find(par, b);
        if (a != a0) atomicMin(&par[a0], a);
        if (b != b0) atomicMin(&par[b0], b);
        if (a == b) return;
        if (a < b) {
            const int t = a;
            a = b;
            b = t;
        }
        const int old = atomicMin(&par[a], b);
        if (old == a) return;
        a = old;
    }
}

__global__ void __launch_bounds__(256)
k_ccl_tile(const int32_t *__restrict__ labels, int H, int W, int32_t *__restrict__ parent, int32_t *__restrict__ csize,
           int32_t *__restrict__ ymax_g, int32_t *__restrict__ xmin_g, int32_t *__restrict__ xmax_g,
           int32_t *__restrict__ lroots, int32_t *__restrict__ lsize, int32_t *__restrict__ lbox, int32_t *__restrict__ ntile,
           int32_t *counters, size_t zs)
{
    ZSHIFT(labels, zs); ZSHIFT(parent, zs); ZSHIFT(csize, zs); ZSHIFT(ymax_g, zs); ZSHIFT(xmin_g, zs); ZSHIFT(xmax_g, zs);
    ZSHIFT(lroots, zs); ZSHIFT(lsize, zs); ZSHIFT(lbox, zs); ZSHIFT(ntile, zs); ZSHIFT(counters, zs);
    __shared__ int slab[CT_H * CT_W];      // labels; after the unions: slot of a local root
    __shared__ int spar[CT_H * CT_W];
    __shared__ int c_root[CT_SLOTS], c_size[CT_SLOTS], c_ymax[CT_SLOTS], c_xmin[CT_SLOTS], c_xmax[CT_SLOTS];
    __shared__ int s_n;
    constexpr int OUT = (int)0x80000000;   // outside the image (never a label)
    constexpr int RPW = CT_H / 4;          // rows per wave
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tx0 = blockIdx.x * CT_W, ty0 = blockIdx.y * CT_H;
    const int x = tx0 + lane;
    const unsigned long long le = (lane == 63) ? ~0ULL : ((2ULL << lane) - 1ULL);      // lanes <= me
    if (tid == 0) s_n = 0;

    int l[RPW];
    bool cont[RPW];
    unsigned long long starts[RPW];
#pragma unroll
    for (int j = 0; j < RPW; ++j) {
        const int y = ty0 + wave * RPW + j;
        l[j] = (x < W && y < H) ? labels[(size_t)y * W + x] : OUT;
    }
#pragma unroll
    for (int j = 0; j < RPW; ++j) {
        const int idx = (wave * RPW + j) * CT_W + lane;
        const int left = __shfl_up(l[j], 1, 64);
        const bool active = l[j] != OUT;
        cont[j] = lane > 0 && active && left == l[j];
        starts[j] = __ballot(active && !cont[j]);
        const unsigned long long below = starts[j] & le;
        const int start_lane = below ? 63 - __clzll((long long)below) : lane;
        spar[idx] = active ? idx - (lane - start_lane) : idx;
        slab[idx] = l[j];
    }
    __syncthreads();
    // vertical: one union per pair of overlapping runs (the pair further left does it when it joins the same two runs)
#pragma unroll
    for (int j = 0; j < RPW; ++j) {
        const int row = wave * RPW + j;
        if (row == 0) continue;                                        // wave-uniform
        const int idx = row * CT_W + lane;
        const int up = j > 0 ? l[j - 1] : slab[idx - CT_W];
        const int upleft = __shfl_up(up, 1, 64);
        if (l[j] != OUT && up == l[j] && !(cont[j] && upleft == l[j])) lds_union(spar, idx, idx - CT_W);
    }
    __syncthreads();
    // roots of the eight pixels of this lane by pointer jumping, all eight chains in flight at once
    int rj[RPW];
#pragma unroll
    for (int j = 0; j < RPW; ++j) rj[j] = spar[(wave * RPW + j) * CT_W + lane];
    while (true) {
        int nx[RPW];
        bool moved = false;
#pragma unroll
        for (int j = 0; j < RPW; ++j) nx[j] = spar[rj[j]];
#pragma unroll
        for (int j = 0; j < RPW; ++j) {
            moved |= nx[j] != rj[j];
            rj[j] = nx[j];
        }
        if (!__any(moved)) break;
    }
#pragma unroll
    for (int j = 0; j < RPW; ++j) {
        const int idx = (wave * RPW + j) * CT_W + lane;
        if (l[j] == OUT) rj[j] = -1;
        if (rj[j] == idx) {
            const int sl = atomicAdd(&s_n, 1);
            if (sl < CT_SLOTS) {
                c_root[sl] = idx;
                c_size[sl] = 0;
                c_ymax[sl] = 0;
                c_xmin[sl] = CT_W - 1;
                c_xmax[sl] = 0;
                slab[idx] = sl;
            }
        }
    }
    __syncthreads();
    const int ns = s_n;
    const int tile = blockIdx.y * gridDim.x + blockIdx.x;
    if (ns > CT_SLOTS) {                                               // block-uniform
        if (tid == 0) {
            atomicOr(&counters[CNT_FLAG], 1);
            ntile[tile] = 0;
        }
        return;
    }
    // size and bounding box of the local components: one set of LDS atomics per horizontal run
#pragma unroll
    for (int j = 0; j < RPW; ++j) {
        const int nact = __popcll(__ballot(l[j] != OUT));
        if (l[j] != OUT && !cont[j]) {
            const unsigned long long above = starts[j] & ~le;
            const int end = above ? __ffsll((long long)above) - 1 : nact;      // one past the run
            const int sl = slab[rj[j]];
            atomicAdd(&c_size[sl], end - lane);
            atomicMax(&c_ymax[sl], wave * RPW + j);
            atomicMin(&c_xmin[sl], lane);
            atomicMax(&c_xmax[sl], end - 1);
        }
    }
    if (tid == 0) ntile[tile] = ns;
    __syncthreads();
    if (tid < ns) {
        const int r = c_root[tid];
        const int g = (ty0 + (r >> 6)) * W + tx0 + (r & 63);
        csize[g] = 0;
        ymax_g[g] = -1;
        xmin_g[g] = 0x7fffffff;
        xmax_g[g] = -1;
        const int pos = tile * CT_SLOTS + tid;
        lroots[pos] = g;
        lsize[pos] = c_size[tid];
        lbox[pos] = (c_ymax[tid] << 16) | (c_xmin[tid] << 8) | c_xmax[tid];
    }
#pragma unroll
    for (int j = 0; j < RPW; ++j) {
        const int y = ty0 + wave * RPW + j;
        if (l[j] != OUT) parent[(size_t)y * W + x] = (ty0 + (rj[j] >> 6)) * W + tx0 + (rj[j] & 63);
    }
}

__global__ void __launch_bounds__(256)
k_ccl_border(const int32_t *__restrict__ labels, int H, int W, int32_t *parent, const int32_t *__restrict__ counters, size_t zs)
{
    ZSHIFT(labels, zs); ZSHIFT(parent, zs); ZSHIFT(counters, zs);
    if (counters[CNT_FLAG]) return;                  // a tile gave up: parent[] is incomplete, the general path follows
    long t = (long)blockIdx.x * 256 + threadIdx.x;
    const int nby = (H - 1) / CT_H, nbx = (W - 1) / CT_W;
    if (t < (long)nby * W) {
        const int y = (int)(t / W + 1) * CT_H, x = (int)(t % W);
        const int p = y * W + x, l = labels[p];
        if (labels[p - W] != l) return;
        if ((x % CT_W) != 0 && labels[p - 1] == l && labels[p - W - 1] == l) return;
        uf_union(parent, p, p - W);
        return;
    }
    t -= (long)nby * W;
    if (t < (long)nbx * H) {
        const int x = (int)(t / H + 1) * CT_W, y = (int)(t % H);
        const int p = y * W + x, l = labels[p];
        if (labels[p - 1] != l) return;
        if ((y % CT_H) != 0 && labels[p - W] == l && labels[p - W - 1] == l) return;
        uf_union(parent, p, p - 1);
    }
}

__global__ void __launch_bounds__(256)
k_lroot_merge(const int32_t *__restrict__ lroots, const int32_t *__restrict__ lsize, const int32_t *__restrict__ lbox,
              const int32_t *__restrict__ ntile, int n_slots, const int32_t *__restrict__ counters, int32_t *parent, int32_t *csize, int32_t *ymax_g, int32_t *xmin_g,
              int32_t *xmax_g, int W, size_t zs)
{
    ZSHIFT(lroots, zs); ZSHIFT(lsize, zs); ZSHIFT(lbox, zs); ZSHIFT(ntile, zs); ZSHIFT(counters, zs); ZSHIFT(parent, zs); ZSHIFT(csize, zs);
    ZSHIFT(ymax_g, zs); ZSHIFT(xmin_g, zs); ZSHIFT(xmax_g, zs);
    if (counters[CNT_FLAG]) return;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n_slots || (i & (CT_SLOTS - 1)) >= ntile[i / CT_SLOTS]) return;
    const int r = lroots[i];
    const int g = uf_find(parent, r);
    if (g != r) parent[r] = g;
    atomicAdd(&csize[g], lsize[i]);
    const int b = lbox[i];
    const int ty0 = (r / W) & ~(CT_H - 1), tx0 = (r % W) & ~(CT_W - 1);
    atomicMax(&ymax_g[g], ty0 + (b >> 16));
    atomicMin(&xmin_g[g], tx0 + ((b >> 8) & 255));
    atomicMax(&xmax_g[g], tx0 + (b & 255));
}

// (positions inside the two lists: counted in LDS per workgroup of 16 tiles, one global atomic per list and workgroup --
// a few thousand single-lane atomics on one address serialise at ~6 ns each)
__global__ void __launch_bounds__(1024)
k_root_classify(const int32_t *__restrict__ lroots, const int32_t *__restrict__ ntile, int n_slots, int32_t *counters,
                const int32_t *__restrict__ parent, const int32_t *__restrict__ csize, int min_size, int max_size, int32_t *kept,
                int32_t *list, size_t zs)
{
    ZSHIFT(lroots, zs); ZSHIFT(ntile, zs); ZSHIFT(counters, zs); ZSHIFT(parent, zs); ZSHIFT(csize, zs); ZSHIFT(kept, zs); ZSHIFT(list, zs);
    __shared__ int n_k, n_s, base_k, base_s;
    if (counters[CNT_FLAG]) return;
    if (threadIdx.x == 0) n_k = n_s = 0;
    __syncthreads();
    const int i = blockIdx.x * 1024 + threadIdx.x;
    int r = -1, kind = 0, pos = 0;                     // kind 1 kept, 2 small
    if (i < n_slots && (i & (CT_SLOTS - 1)) < ntile[i / CT_SLOTS]) {
        r = lroots[i];
        if (parent[r] == r) {
            const int sz = csize[r];
            if (sz >= max_size) atomicAdd(&counters[CNT_OVER], 1);
            else if (sz >= min_size) { kind = 1; pos = atomicAdd(&n_k, 1); }
            else { kind = 2; pos = atomicAdd(&n_s, 1); }
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        base_k = n_k ? atomicAdd(&counters[CNT_KEPT], n_k) : 0;
        base_s = n_s ? atomicAdd(&counters[CNT_SMALL], n_s) : 0;
    }
    __syncthreads();
    if (kind == 1 && base_k + pos < CONN_KEPT_CAP) kept[base_k + pos] = r;
    if (kind == 2) list[base_s + pos] = r;
}

// consecutive labels of the kept components = raster rank of their roots: bucket sort by pixel index in LDS, by one
// workgroup of NT threads (the last workgroup of the k_small_bfs_reg launch: it needs nothing from the others and
// hides behind them)
constexpr int KR_BUCKETS = 1024, KR_SORTED_LDS = 2048;
template <int NT>
__device__ __forceinline__ void kept_rank_block(int *offs /* [KR_BUCKETS + 1] */, int *cursor /* [KR_BUCKETS] */,
                                                int *wsum /* [NT / 64] */, int *sorted_lds /* [KR_SORTED_LDS] */,
                                                const int32_t *__restrict__ kept, const int32_t *__restrict__ counters,
                                                long n_pixels, int start_label, int32_t *sorted_global, int32_t *newlabel)
{
    const int tid = threadIdx.x;
    const int nk = counters[CNT_KEPT];
    if (nk > CONN_KEPT_CAP || counters[CNT_FLAG]) return;                  // the host sees the count and takes the general path
    int *sorted = nk <= KR_SORTED_LDS ? sorted_lds : sorted_global;
    for (int b = tid; b < KR_BUCKETS; b += NT) cursor[b] = 0;
    __syncthreads();
    for (int k = tid; k < nk; k += NT) atomicAdd(&cursor[(int)(((long)kept[k] * KR_BUCKETS) / n_pixels)], 1);
    __syncthreads();
    {   // exclusive scan of the bucket counts: KR_BUCKETS / NT consecutive buckets per thread
        constexpr int PER = KR_BUCKETS / NT;
        int v[PER], sum = 0;
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            v[j] = cursor[tid * PER + j];
            sum += v[j];
        }
        const int lane = tid & 63, wave = tid >> 6;
        int incl = sum;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int t = __shfl_up(incl, off, 64);
            if (lane >= off) incl += t;
        }
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        int base = 0;
        for (int w = 0; w < wave; ++w) base += wsum[w];
        int run = base + incl - sum;
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            offs[tid * PER + j] = run;
            run += v[j];
        }
        if (tid == NT - 1) offs[KR_BUCKETS] = run;
        __syncthreads();
        for (int b = tid; b < KR_BUCKETS; b += NT) cursor[b] = offs[b];
    }
    __syncthreads();
    for (int k = tid; k < nk; k += NT) {
        const int r = kept[k];
        sorted[atomicAdd(&cursor[(int)(((long)r * KR_BUCKETS) / n_pixels)], 1)] = r;
    }
    __threadfence();
    __syncthreads();
    for (int i = tid; i < nk; i += NT) {
        const int r = sorted[i];
        const int b = (int)(((long)r * KR_BUCKETS) / n_pixels);
        int rank = offs[b];
        for (int j = offs[b]; j < offs[b + 1]; ++j) rank += sorted[j] < r;
        newlabel[r] = start_label + rank;
    }
}

// `adjacent` of the small components (2-D): exact emulation of the reference's BFS (queue order, neighbour order
// +x, -x, +y, -y) by one wave with the frontier in registers -- lane i holds the cell of BFS rank i inside the current
// level.  The bounding box of the component plus two rings is staged in LDS by the whole workgroup as one byte per
// cell: 0 other, 1 member not yet discovered, 2 pixel of an earlier component (smaller root: already labelled when the
// reference gets here), 3 member done, 4 + i member of the current frontier with rank i.  A frontier cell claims the
// undiscovered neighbour n unless one of the other three neighbours of n is a frontier cell of lower rank (that one
// discovers it first); the claims of a level, ordered by (rank, direction), are the next frontier: positions from four
// ballots.  `adjacent` is the earlier-component neighbour met by the highest (level, rank, direction).
constexpr int BR_CELLS = 16384;      // 16.6 KB of LDS per workgroup: every small component of an image is resident at once
constexpr int BR_CELLS_BIG = 131072; // second launch (a few workgroups, 128 KB of LDS each) for the bounding boxes beyond that:
                                     // a one-pixel sliver along a 300-pixel diagonal edge has 143 pixels in 10^5 cells
// list / count: the components to do (count capped at count_cap); rej_*: where the ones this launch cannot take go
// (with their tight bounding box, for k_small_bfs_wave); cells_cap: cells of the LDS tile (dynamic shared memory =
// cells_cap + 256 bytes); with_rank: the last workgroup ranks the kept roots instead
__global__ void __launch_bounds__(256)
k_small_bfs_reg(const int32_t *__restrict__ list, const int32_t *__restrict__ count, int count_cap, int32_t *counters,
                const int32_t *__restrict__ parent, const int32_t *__restrict__ ymax_g, const int32_t *__restrict__ xmin_g,
                const int32_t *__restrict__ xmax_g, int H, int W, int32_t *adjptr, int32_t *rej_list, int rej_counter,
                int32_t *rej_bbox, int cells_cap, int with_rank, const int32_t *__restrict__ kept, int start_label,
                int32_t *sorted, int32_t *newlabel, size_t zs)
{
    ZSHIFT(list, zs); ZSHIFT(count, zs); ZSHIFT(counters, zs); ZSHIFT(parent, zs); ZSHIFT(ymax_g, zs); ZSHIFT(xmin_g, zs); ZSHIFT(xmax_g, zs);
    ZSHIFT(adjptr, zs); ZSHIFT(rej_list, zs); ZSHIFT(rej_bbox, zs); ZSHIFT(kept, zs); ZSHIFT(sorted, zs); ZSHIFT(newlabel, zs);
    static_assert(BR_CELLS + 64 * 4 >= (2 * KR_BUCKETS + 1 + 4 + KR_SORTED_LDS) * 4, "shared memory of the two roles");
    extern __shared__ __align__(16) uint8_t smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    int n_workers = gridDim.x;
    if (with_rank) {
        n_workers -= 1;
        if (blockIdx.x == gridDim.x - 1) {
            int *si = reinterpret_cast<int *>(smem);
            kept_rank_block<256>(si, si + KR_BUCKETS + 1, si + 2 * KR_BUCKETS + 1, si + 2 * KR_BUCKETS + 1 + 4, kept, counters,
                                 (long)H * W, start_label, sorted, newlabel);
            return;
        }
    }
    uint8_t *cs = smem;
    int *fr = reinterpret_cast<int *>(smem + cells_cap);
    const int n_small = counters[CNT_FLAG] ? 0 : min(*count, count_cap);
    for (int t = blockIdx.x; t < n_small; t += n_workers) {
        const int root = list[t];
        const int ry = root / W, rx = root - ry * W;
        const int y1 = ymax_g[root], x0 = xmin_g[root], x1 = xmax_g[root];
        const int by0 = ry - 2, bx0 = x0 - 2;
        const int bw = x1 - x0 + 5, bh = y1 - ry + 5;
        const long cells_l = (long)bw * bh;
        bool reject = cells_l > cells_cap;                // block-uniform
        int adj_cell = -1;
        if (!reject) {
            const int cells = (int)cells_l;
            const unsigned int m_bw = (unsigned int)((0x100000000ull + bw - 1) / bw);
            __syncthreads();                              // the previous component is done with the tile
            // four cells per thread and round: the two dependent loads of four cells are in flight together
            for (int c0 = tid; c0 < cells; c0 += 4 * 256) {
                int q[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int c = c0 + u * 256;
                    // (the reciprocal is exact while c * bw < 2^32: always for the 16 K tile)
                    const int cy = cells_cap <= BR_CELLS ? (int)__umulhi((unsigned int)c, m_bw) : c / bw, cx = c - cy * bw;
                    const int y = by0 + cy, x = bx0 + cx;
                    q[u] = (c < cells && y >= 0 && y < H && x >= 0 && x < W) ? parent[(size_t)y * W + x] : -1;
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) q[u] = q[u] >= 0 ? parent[q[u]] : 0x7fffffff;
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (c0 + u * 256 < cells) cs[c0 + u * 256] = (uint8_t)(q[u] == root ? 1 : (q[u] < root ? 2 : 0));
            }
            __syncthreads();
            if (tid < 64) {
                int c = 2 * bw + (rx - bx0);              // the root: row 2 of the tile
                int f = 1;
                if (lane == 0) cs[c] = 4;
                const unsigned long long lt = (1ULL << lane) - 1ULL;
                // (one wave: its LDS operations execute in program order; the compiler barriers keep the reads of a level
                // behind the writes of the previous one)
                while (f > 0) {
                    __asm__ volatile("" ::: "memory");
                    if (f == 1) {
                        // one frontier cell (every level of a one-pixel-wide sliver): nothing competes, everything is
                        // wave-uniform -- one LDS round trip, no ballots, no exchange through LDS
                        const int cu = __builtin_amdgcn_readfirstlane(c);
                        const int u0 = __builtin_amdgcn_readfirstlane((int)cs[cu + 1]);
                        const int u1 = __builtin_amdgcn_readfirstlane((int)cs[cu - 1]);
                        const int u2 = __builtin_amdgcn_readfirstlane((int)cs[cu + bw]);
                        const int u3 = __builtin_amdgcn_readfirstlane((int)cs[cu - bw]);
                        if (u3 == 2) adj_cell = cu - bw;
                        else if (u2 == 2) adj_cell = cu + bw;
                        else if (u1 == 2) adj_cell = cu - 1;
                        else if (u0 == 2) adj_cell = cu + 1;
                        int k = 0;
                        __asm__ volatile("" ::: "memory");
                        if (lane == 0) cs[cu] = 3;
                        if (u0 == 1) { if (lane == k) { c = cu + 1; cs[c] = (uint8_t)(4 + k); } ++k; }
                        if (u1 == 1) { if (lane == k) { c = cu - 1; cs[c] = (uint8_t)(4 + k); } ++k; }
                        if (u2 == 1) { if (lane == k) { c = cu + bw; cs[c] = (uint8_t)(4 + k); } ++k; }
                        if (u3 == 1) { if (lane == k) { c = cu - bw; cs[c] = (uint8_t)(4 + k); } ++k; }
                        f = k;
                        continue;
                    }
                    const bool act = lane < f;
                    if (!act) c = 2 * bw + 2;             // any cell two rings inside: keeps the addresses valid
                    const int v0 = cs[c + 1], v1 = cs[c - 1], v2 = cs[c + bw], v3 = cs[c - bw];
                    const int e0 = cs[c + 2], e1 = cs[c - 2], e2 = cs[c + 2 * bw], e3 = cs[c - 2 * bw];
                    const int epp = cs[c + 1 + bw], epm = cs[c + 1 - bw], emp = cs[c - 1 + bw], emm = cs[c - 1 - bw];
                    auto lower = [lane](int v) { return (unsigned int)(v - 4) < (unsigned int)lane; };
                    const bool w0 = act && v0 == 1 && !(lower(e0) || lower(epp) || lower(epm));
                    const bool w1 = act && v1 == 1 && !(lower(e1) || lower(emp) || lower(emm));
                    const bool w2 = act && v2 == 1 && !(lower(e2) || lower(epp) || lower(emp));
                    const bool w3 = act && v3 == 1 && !(lower(e3) || lower(epm) || lower(emm));
                    const unsigned long long con = __ballot(act && (v0 == 2 || v1 == 2 || v2 == 2 || v3 == 2));
                    if (con) {
                        const int cand = v3 == 2 ? c - bw : (v2 == 2 ? c + bw : (v1 == 2 ? c - 1 : c + 1));
                        adj_cell = __builtin_amdgcn_readlane(cand, 63 - __clzll((long long)con));
                    }
                    const unsigned long long b0 = __ballot(w0), b1 = __ballot(w1), b2 = __ballot(w2), b3 = __ballot(w3);
                    const int total = __popcll(b0) + __popcll(b1) + __popcll(b2) + __popcll(b3);
                    if (total > 64) {
                        reject = true;
                        break;
                    }
                    const int p0 = __popcll(b0 & lt) + __popcll(b1 & lt) + __popcll(b2 & lt) + __popcll(b3 & lt);
                    const int p1 = p0 + (w0 ? 1 : 0), p2 = p1 + (w1 ? 1 : 0), p3 = p2 + (w2 ? 1 : 0);
                    __asm__ volatile("" ::: "memory");
                    if (act) cs[c] = 3;
                    if (w0) { fr[p0] = c + 1; cs[c + 1] = (uint8_t)(4 + p0); }
                    if (w1) { fr[p1] = c - 1; cs[c - 1] = (uint8_t)(4 + p1); }
                    if (w2) { fr[p2] = c + bw; cs[c + bw] = (uint8_t)(4 + p2); }
                    if (w3) { fr[p3] = c - bw; cs[c - bw] = (uint8_t)(4 + p3); }
                    f = total;
                    __asm__ volatile("" ::: "memory");
                    c = fr[lane];
                }
            }
        }
        if (tid == 0) {
            if (reject) {
                // hand over to the LDS-frontier kernel (k_small_bfs_wave): it gets the tight bounding box
                adjptr[root] = -1;                    // (defined even if nobody takes it: k_lroot_labels walks these)
                const int i = atomicAdd(&counters[rej_counter], 1);
                if (i < CONN_FB_CAP) {
                    rej_list[i] = root;
                    rej_bbox[6 * i + 0] = ry; rej_bbox[6 * i + 1] = y1;
                    rej_bbox[6 * i + 2] = x0; rej_bbox[6 * i + 3] = x1;
                    rej_bbox[6 * i + 4] = 0; rej_bbox[6 * i + 5] = 0;
                }
            } else {
                int adj = -1;
                if (adj_cell >= 0) {
                    const int cy = adj_cell / bw, cx = adj_cell - cy * bw;
                    adj = parent[parent[(size_t)(by0 + cy) * W + bx0 + cx]];
                }
                adjptr[root] = adj;
            }
        }
    }
}

// labels of all tile-local roots (what k_write_labels looks up): kept component -> its consecutive label; small
// component -> the label of the kept component at the end of its `adjacent` chain (roots strictly decrease), 0 when the
// BFS met no labelled neighbour (k_small_resolve of the general path, done per local root)
__global__ void __launch_bounds__(256)
k_lroot_labels(const int32_t *__restrict__ lroots, const int32_t *__restrict__ ntile, int n_slots,
               const int32_t *__restrict__ counters, const int32_t *__restrict__ parent,
               const int32_t *__restrict__ csize, const int32_t *__restrict__ adjptr, int min_size, int32_t *newlabel, size_t zs)
{
    ZSHIFT(lroots, zs); ZSHIFT(ntile, zs); ZSHIFT(counters, zs); ZSHIFT(parent, zs); ZSHIFT(csize, zs); ZSHIFT(adjptr, zs); ZSHIFT(newlabel, zs);
    if (counters[CNT_FLAG]) return;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n_slots || (i & (CT_SLOTS - 1)) >= ntile[i / CT_SLOTS]) return;
    const int r = lroots[i];
    int g = parent[r];
    const bool kept_root = g == r && csize[g] >= min_size;
    if (kept_root) return;                             // (its label comes from the ranking)
    while (g >= 0 && csize[g] < min_size) g = adjptr[g];
    newlabel[r] = g >= 0 ? newlabel[g] : 0;
}

// Enqueues the tile path for zb.nz maps of one size (image b: every pointer b * zb.zs bytes further on) and ends with the ONE
// host synchronisation of the stage; ok[b] = the fast path produced the result of image b (labels_out, n_kept[b]).
static int conn_fast_2d(const int32_t *labels_in, int H, int W, int min_size, int max_size, int start_label, const ConnWork &w,
                        int32_t *labels_out, int *n_kept, bool *ok, hipStream_t st, ZBatch zb = ZBatch(), int32_t *batch_stage = nullptr)
{
    if (zb.nz > 1 && !batch_stage) {
        set_error("connectivity: a batch needs a staging area for its counters");
        return -1;
    }
    const int n = H * W;
    const unsigned nz = (unsigned)zb.nz;
    const size_t zs = zb.zs;
    const dim3 tiles(cdiv(W, CT_W), cdiv(H, CT_H), nz);
    const int n_tiles = tiles.x * tiles.y, n_slots = n_tiles * CT_SLOTS;
    if (CONN_DENSE_INTS + (size_t)n_tiles * (3 * CT_SLOTS + 1) > w.dense_ints) {
        for (unsigned b = 0; b < nz; ++b) ok[b] = false;  // scratch sized without the per-tile lists: the general path takes it
        return 0;
    }
    const ConnDense d = conn_dense(w, n_tiles);
    int32_t *ymax_g = w.queue, *xmin_g = w.slotmap, *xmax_g = w.bbox;
    std::vector<int32_t> host_counters((size_t)16 * nz);
    if (launch_zero(w.counters, 16 * sizeof(int32_t), st, zb)) return -1;
    hipLaunchKernelGGL(k_ccl_tile, tiles, 256, 0, st, labels_in, H, W, w.parent, w.csize, ymax_g, xmin_g, xmax_g, d.lroots, d.lsize,
                       d.lbox, d.ntile, w.counters, zs);
    const long nborder = (long)((H - 1) / CT_H) * W + (long)((W - 1) / CT_W) * H;
    if (nborder > 0)
        hipLaunchKernelGGL(k_ccl_border, dim3(cdiv(nborder, 256), 1, nz), 256, 0, st, labels_in, H, W, w.parent, w.counters, zs);
    const dim3 lgrid(cdiv(n_slots, 256), 1, nz);
    hipLaunchKernelGGL(k_lroot_merge, lgrid, 256, 0, st, d.lroots, d.lsize, d.lbox, d.ntile, n_slots, w.counters, w.parent, w.csize, ymax_g,
                       xmin_g, xmax_g, W, zs);
    hipLaunchKernelGGL(k_root_classify, dim3(cdiv(n_slots, 1024), 1, nz), 1024, 0, st, d.lroots, d.ntile, n_slots, w.counters, w.parent, w.csize,
                       min_size, max_size, d.kept, w.list, zs);
    // (a batch: the images are much smaller than the 2048^2 the 2 048 workgroups were sized for -- a workgroup takes the small
    // components i, i + grid, ... of its image)
    const int bfs_blocks = nz > 1 ? std::max(64, 2048 / (int)nz) : 2048;
    hipLaunchKernelGGL(k_small_bfs_reg, dim3(bfs_blocks + 1, 1, nz), 256, BR_CELLS + 256, st, w.list, w.counters + CNT_SMALL, n, w.counters, w.parent,
                       ymax_g, xmin_g, xmax_g, H, W, w.adjptr, d.fb_list, (int)CNT_FB, d.fb_bbox, BR_CELLS, 1, d.kept, start_label,
                       d.sorted, w.newlabel, zs);
    // bounding boxes beyond the 16 K tile (long thin slivers): the same kernel with a 128 K tile, a few workgroups
    {
        static bool big_attr[IMSEGM_MAX_DEVICES];
        int dev = 0;
        HIP_TRY(hipGetDevice(&dev));
        if (dev < 0 || dev >= IMSEGM_MAX_DEVICES || !big_attr[dev]) {
            HIP_TRY(hipFuncSetAttribute((const void *)k_small_bfs_reg, hipFuncAttributeMaxDynamicSharedMemorySize, BR_CELLS_BIG + 256));
            if (dev >= 0 && dev < IMSEGM_MAX_DEVICES) big_attr[dev] = true;
        }
    }
    hipLaunchKernelGGL(k_small_bfs_reg, dim3(nz > 1 ? 32 : 128, 1, nz), 256, BR_CELLS_BIG + 256, st, d.fb_list, w.counters + CNT_FB, CONN_FB_CAP,
                       w.counters, w.parent, ymax_g, xmin_g, xmax_g, H, W, w.adjptr, d.fb2_list, (int)CNT_FB2, d.fb2_bbox, BR_CELLS_BIG, 0,
                       d.kept, start_label, d.sorted, w.newlabel, zs);
    // frontiers of more than 64 cells: the LDS-frontier kernel; what this one cannot take either ends up in CNT_FALLBACK and
    // sends the image to the general path
    hipLaunchKernelGGL((k_small_bfs_wave<8192, 2048, true>), dim3(nz > 1 ? 64 : 256, 1, nz), 64, 0, st, d.fb2_list, w.counters, w.counters + CNT_FB2,
                       w.parent, d.fb2_bbox, 1, H, W, (const int32_t *)nullptr, CONN_FB_CAP, w.adjptr, d.fb_reject, zs);
    hipLaunchKernelGGL(k_lroot_labels, lgrid, 256, 0, st, d.lroots, d.ntile, n_slots, w.counters, w.parent, w.csize, w.adjptr, min_size,
                       w.newlabel, zs);
    hipLaunchKernelGGL(k_write_labels, dim3(cdiv(cdiv(n, 4), 256), 1, nz), 256, 0, st, w.parent, w.newlabel, n, labels_out,
                       (const int32_t *)(w.counters + CNT_FLAG), zs);
    HIP_TRY(hipGetLastError());
    if (nz == 1) {
        HIP_TRY(hipMemcpyAsync(host_counters.data(), w.counters, 16 * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    } else {
        // the counter blocks of the images side by side behind the last image's (its `dense` lists are through), one transfer
        int32_t *stage = batch_stage;
        if (launch_copy_rows(stage, 16 * sizeof(int32_t), w.counters, zs, 16 * sizeof(int32_t), (int)nz, st)) return -1;
        HIP_TRY(hipMemcpyAsync(host_counters.data(), stage, (size_t)nz * 16 * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    }
    HIP_TRY(hipStreamSynchronize(st));
    for (unsigned b = 0; b < nz; ++b) {
        const int32_t *hc = host_counters.data() + (size_t)16 * b;
        ok[b] = hc[CNT_FLAG] == 0 && hc[CNT_OVER] == 0 && hc[CNT_FALLBACK] == 0 && hc[CNT_KEPT] <= CONN_KEPT_CAP &&
                hc[CNT_FB] <= CONN_FB_CAP && hc[CNT_FB2] <= CONN_FB_CAP;
        n_kept[b] = hc[CNT_KEPT];
    }
    return 0;
}

// The tile path for a batch of label maps of one size (csrc/batch.hip): image b lives b * zb.zs bytes behind every pointer of `w`,
// labels_in and labels_out.  n_labels_out[b] < 0: image b has to go through launch_enforce_connectivity by itself (the general path).
int launch_enforce_connectivity_batch(const int32_t *labels_in, int H, int W, long min_size_l, long max_size_l, int start_label, ConnWork w,
                                      int32_t *labels_out, int *n_labels_out, hipStream_t st, ZBatch zb, int32_t *stage_dev)
{
    const int min_size = (int)std::min<long>(min_size_l, 0x7fffffff);
    const int max_size = (int)std::min<long>(max_size_l, 0x7fffffff);
    std::vector<int> n_kept(zb.nz);
    std::vector<char> ok(zb.nz);
    static_assert(sizeof(bool) == sizeof(char), "ok flags");
    if (conn_fast_2d(labels_in, H, W, min_size, max_size, start_label, w, labels_out, n_kept.data(), reinterpret_cast<bool *>(ok.data()), st, zb,
                     stage_dev))
        return -1;
    for (int b = 0; b < zb.nz; ++b) n_labels_out[b] = ok[b] ? (n_kept[b] > 0 ? start_label + n_kept[b] : 1) : -1;
    return 0;
}

// how often a 2-D map had to take the general path (diagnostic: tests assert that ordinary inputs stay on the tile path)
static std::atomic<long> g_conn_general_runs{0};
long conn_general_runs() { return g_conn_general_runs.load(); }

int launch_enforce_connectivity(const int32_t *labels_in, int D, int H, int W, long min_size_l, long max_size_l,
                                int start_label, ConnWork w, int32_t *labels_out, int *n_labels_out_host,
                                hipStream_t st)
{
    const int n = D * H * W;
    const int grid = cdiv(n, 256);
    const int min_size = (int)std::min<long>(min_size_l, 0x7fffffff);
    const int max_size = (int)std::min<long>(max_size_l, 0x7fffffff);
    uint8_t *state = w.visited + n;       // second half of the byte scratch (2 * n bytes)
    int32_t host_counters[16];

    if (D == 1 && !knobs().conn_general) {
        bool ok = false;
        int n_kept = 0;
        if (conn_fast_2d(labels_in, H, W, min_size, max_size, start_label, w, labels_out, &n_kept, &ok, st)) return -1;
        if (ok) {
            *n_labels_out_host = n_kept > 0 ? start_label + n_kept : 1;
            return 0;
        }
        g_conn_general_runs.fetch_add(1);
    }

    // fast path, speculating that no component reaches max_size: one CCL round, then the tail;
    // a single host synchronisation at the very end reads the counters
    // (every pixel is active in this round: the kernels do not look at the state bytes)
    HIP_TRY(hipMemsetAsync(w.counters, 0, 16 * sizeof(int32_t), st));
    conn_ccl_round<true>(labels_in, D, H, W, max_size, w, state, st);
    if (conn_tail(w.csize, w.adjptr, D, H, W, min_size, start_label, w, labels_out, st, true, max_size)) return -1;
    HIP_TRY(hipMemcpyAsync(host_counters, w.counters, 8 * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));

    if (host_counters[CNT_OVER] > 0) {
        // general path: truncate oversize components to their first max_size pixels in BFS order
        // (exact sequential BFS by one thread each), re-label the left-overs, repeat
        int32_t *csize_final = w.adjptr;      // final sizes are gathered here during the rounds
        HIP_TRY(hipMemsetAsync(state, ST_ACTIVE, n, st));
        for (int round = 0;; ++round) {
            HIP_TRY(hipMemsetAsync(w.visited, 0, n, st));
            HIP_TRY(hipMemsetAsync(w.counters, 0, 3 * sizeof(int32_t), st));
            conn_ccl_round<false>(labels_in, D, H, W, max_size, w, state, st);
            HIP_TRY(hipMemcpyAsync(host_counters, w.counters, 4 * sizeof(int32_t), hipMemcpyDeviceToHost, st));
            HIP_TRY(hipStreamSynchronize(st));
            int n_over = host_counters[CNT_OVER];
            if (n_over > 0) {
                if ((long)n_over * max_size > (long)n) {
                    set_error("enforce_connectivity: internal queue overflow");
                    return -1;
                }
                hipLaunchKernelGGL(k_oversize_bfs, cdiv(n_over, 64), 64, 0, st, w.list, n_over, w.parent, state, D, H, W,
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
        if (conn_tail(csize_final, w.csize, D, H, W, min_size, start_label, w, labels_out, st, false)) return -1;
        HIP_TRY(hipMemcpyAsync(host_counters, w.counters, 8 * sizeof(int32_t), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
    }
    int n_kept = host_counters[CNT_KEPT];
    *n_labels_out_host = n_kept > 0 ? start_label + n_kept : 1;
    return 0;
}

}  // namespace imsegm
