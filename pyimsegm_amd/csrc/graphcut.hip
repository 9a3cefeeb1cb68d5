// graphcut.hip -- alpha-expansion over the superpixel adjacency graph as ONE persistent workgroup:
// label schedule, binary-energy construction, lock-free push-relabel max-flow and the energy test
// all run on the device without host round trips.
//
// Replaces the native boundary `gco.cut_general_graph(edges, edge_weights, unary_cost,
// pairwise_cost, algorithm='expansion', n_iter=-1)` called at
// /root/reference/imsegm/graph_cuts.py:735-744 (gco-wrapper: pyGCO -> GCO-v3 -> BK maxflow).
// Integer energies exactly as pyGCO builds them (see api.hip); binary move energy as
// Kolmogorov's energy.h (add_term1 / add_term2); cut convention of maxflow-v3 `what_segment(i,
// SOURCE)`: a site keeps its label only if it can still reach the sink in the residual graph, which
// makes the result independent of the max-flow algorithm (oracle: gco_alpha_expansion).
//
// Max-flow: Hong & He's lock-free push-relabel (one owner thread per node, atomics on residual
// capacities / excesses) with periodic global relabelling (reverse BFS from the sink).  Terminals
// are implicit: excess < 0 == remaining capacity towards the sink.  Only phase 1 (maximum
// preflow) is needed, because the minimal sink side of a min cut is already determined by it.
// The mutable arrays (residual capacities, heights, excesses) live in LDS when they fit
// (K ~ 2e3 nodes, E ~ 6e3 edges -> ~75 KB), otherwise in a global scratch buffer.
#include "slic.h"
#include <cstdio>
#include <cstdlib>
#include <atomic>
#include <mutex>

namespace imsegm {

constexpr int GC_THREADS = 1024;         // upper bound; small graphs run with fewer waves (cheaper barriers)
constexpr int GC_MAX_LABELS = 64;
constexpr int GC_PUSH_ROUNDS = 24;        // lock-free push/relabel sweeps between global relabels

struct GcDevice {
    int K, C, E, n_iter;
    const int32_t *E_dev;      // when set: the real number of edges lives on the device, E is its capacity
    const int32_t *edges;      // [E][2]
    const int32_t *w;          // [E]
    const int32_t *unary;      // [K][C]
    const int32_t *smooth;     // [C][C]
    const int32_t *arc_start;  // [K+1] CSR over directed arcs
    const int32_t *arc_to;     // [2E]
    const int32_t *arc_rev;    // [2E] index of the reverse arc
    const int32_t *edge_arc;   // [E][2] arc index of a->b and of b->a
    int32_t *labels;           // [K] in/out (starts at 0 = GCO default labelling)
    int32_t *prop;             // [K] proposed labelling
    long long *energy_out;     // [1]
    int32_t *g_cap;            // [2E] global fallbacks
    int32_t *g_height;         // [K]
    long long *g_excess;       // [K]
    int use_lds;
    int lds_lab;               // labels, proposal and unary costs are copied into LDS as well (written back at the end)
    int lds_topo;              // 1: arc_start + arc_to in LDS, 2: + arc_rev
    int e_cap;                 // edge capacity the LDS layout was sized for (E on the device may be smaller)
    int skip_repeat;           // the smoothness term is a metric: a move that repeats the last accepted label is skipped
    const int32_t *K_dev;      // when set: the real number of sites lives on the device, K is its upper bound (a batch: per image)
    size_t zs;                 // several graphs per launch (ZBatch): graph blockIdx.z, every buffer zs bytes further on per graph
    int32_t *status;           // [1] 0 ok, 1 = max-flow iteration cap hit
    int test_absent;           // (IMSEGM_GC_GRID_TEST_ABSENT) the last workgroup of the grid-wide kernel leaves at once: a block that is not resident
    long long *dbg;            // (IMSEGM_GC_DEBUG) [16] counters / 100 MHz clock sums of thread 0, or null
};
#define GC_DBG_ADD(j, v)                                                                           \
    if (g.dbg && threadIdx.x == 0) g.dbg[j] += (v);
#define GC_DBG_CLOCK(j)                                                                            \
    if (g.dbg && threadIdx.x == 0) {                                                               \
        const long long now_ = (long long)wall_clock64();                                          \
        g.dbg[j] += now_ - g.dbg[15];                                                              \
        g.dbg[15] = now_;                                                                          \
    }

// Accessors for the arrays that other threads modify with atomics.  In the LDS case the scope is
// irrelevant; in the global-scratch case agent scope keeps the loads out of the (non-coherent for
// atomics) vector L1.
__device__ __forceinline__ int ld(const int *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ long long ld(const long long *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st(int *p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st(long long *p, long long v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__device__ __forceinline__ long long block_sum_i64(long long v, long long *scratch)
{
    v = wave_sum_i64(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = v;
    __syncthreads();
    long long t = 0;
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) t += scratch[i];
    return t;
}

__device__ __forceinline__ long long gc_energy(const GcDevice &g, const int32_t *lab, long long *scratch)
{
    long long e = 0;
    for (int i = threadIdx.x; i < g.K; i += blockDim.x) e += g.unary[(size_t)i * g.C + lab[i]];
    for (int j = threadIdx.x; j < g.E; j += blockDim.x) {
        int a = g.edges[2 * j], b = g.edges[2 * j + 1];
        e += (long long)g.w[j] * g.smooth[lab[a] * g.C + lab[b]];
    }
    return block_sum_i64(e, scratch);
}

#ifndef GC_ARCS
#define GC_ARCS 8
#endif

// OR over the workgroup with ONE barrier per call: three LDS words used in turn -- call r sets and reads word r % 3, and thread 0
// clears word (r + 2) % 3 behind the barrier: its last readers (call r - 1) have all arrived at this barrier, its next writers
// (call r + 2) start behind the next one.  (__syncthreads_or costs three barriers; a relabelling level or a push round is little
// more than its barriers.)
__device__ __forceinline__ bool block_or(int pred, int *flags, unsigned &calls)
{
    int *f = flags + calls % 3;
    if (pred) __hip_atomic_store(f, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __syncthreads();
    const int any = __hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (threadIdx.x == 0) __hip_atomic_store(flags + (calls + 2) % 3, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    ++calls;
    return any != 0;
}

// The graph never changes during the kernel: with at most NPT nodes per thread (u = tid + s * blockDim) the arc range of a node and the
// heads and reverse arcs of its first GC_ARCS arcs stay in registers, and a node's pass of a relabelling level or a push round is ONE
// round of independent LDS reads (capacities and the heights of the heads) instead of a chain arc range -> head -> height.
template <int NPT> struct GcTopo {
    int a0[NPT > 0 ? NPT : 1], deg[NPT > 0 ? NPT : 1];
    unsigned tr[NPT > 0 ? NPT : 1][GC_ARCS];          // head | reverse arc << 16 (the launcher checks that both fit)
};

// reverse BFS from the sink over residual arcs: height = exact distance to the sink, HMAX if none
template <int NPT>
__device__ __forceinline__ void gc_global_relabel(const GcDevice &g, const GcTopo<NPT> &t, int *cap, int *height, long long *excess,
                                                  int alpha, int *flags, unsigned &calls)
{
    const int HMAX = g.K + 2;
    unsigned act = 0;                          // bit s: node tid + s * blockDim takes part in the move (its label is not alpha)
    {
        int s = 0;
        for (int u = threadIdx.x; u < g.K; u += blockDim.x, ++s) {
            const bool a = g.labels[u] != alpha;
            if (a && s < 32) act |= 1u << s;
            st(&height[u], (a && ld(&excess[u]) < 0) ? 1 : HMAX);
        }
    }
    __syncthreads();
    for (int level = 1; level < HMAX; ++level) {
        GC_DBG_ADD(2, 1)
        int changed = 0;
        if (NPT > 0) {
#pragma unroll
            for (int s = 0; s < (NPT > 0 ? NPT : 1); ++s) {
                const int u = threadIdx.x + s * blockDim.x;
                if (u >= g.K || !((act >> s) & 1) || t.deg[s] == 0 || ld(&height[u]) != HMAX) continue;
                int c[GC_ARCS], h[GC_ARCS];
#pragma unroll
                for (int i = 0; i < GC_ARCS; ++i) {
                    c[i] = ld(&cap[t.a0[s] + min(i, t.deg[s] - 1)]);
                    h[i] = ld(&height[t.tr[s][i] & 0xffffu]);
                }
                bool hit = false;
#pragma unroll
                for (int i = 0; i < GC_ARCS; ++i) hit |= c[i] > 0 && h[i] == level;
                for (int a = t.a0[s] + GC_ARCS; a < t.a0[s] + t.deg[s] && !hit; ++a)          // (more than GC_ARCS neighbours: rare)
                    hit = ld(&cap[a]) > 0 && ld(&height[g.arc_to[a]]) == level;
                if (hit) {
                    st(&height[u], level + 1);
                    changed = 1;
                }
            }
        } else {
            for (int u = threadIdx.x; u < g.K; u += blockDim.x) {
                if (ld(&height[u]) != HMAX || g.labels[u] == alpha) continue;
                // GC_ARCS arcs at a time, their loads issued together: a node's pass is three dependent reads (arc range -> capacity
                // and head -> height of the head) per batch, not three per arc
                const int a0 = g.arc_start[u], a1 = g.arc_start[u + 1];
                bool hit = false;
                for (int base = a0; base < a1 && !hit; base += GC_ARCS) {
                    int c[GC_ARCS], h[GC_ARCS];
#pragma unroll
                    for (int i = 0; i < GC_ARCS; ++i) {
                        const int a = min(base + i, a1 - 1);
                        c[i] = ld(&cap[a]);
                        h[i] = g.arc_to[a];
                    }
#pragma unroll
                    for (int i = 0; i < GC_ARCS; ++i) h[i] = ld(&height[h[i]]);
#pragma unroll
                    for (int i = 0; i < GC_ARCS; ++i) hit |= c[i] > 0 && h[i] == level;
                }
                if (hit) {
                    st(&height[u], level + 1);
                    changed = 1;
                }
            }
        }
        // (a node that got level + 1 in this pass is not read as `level` by anybody: one barrier with the OR is enough)
        if (!block_or(changed, flags, calls)) break;
    }
}

// one expansion move; returns (uniformly) whether the energy strictly decreased
template <int NPT>
__device__ __forceinline__ bool gc_expand(const GcDevice &g, const GcTopo<NPT> &t, int alpha, int *cap, int *height, long long *excess,
                                          long long *energy, int *flags, unsigned &calls, long long *scratch)
{
    const int HMAX = g.K + 2;
    // any active site at all?
    int mine = 0;
    for (int u = threadIdx.x; u < g.K; u += blockDim.x) {
        int l = g.labels[u];
        st(&excess[u], (l != alpha) ? (long long)g.unary[(size_t)u * g.C + l] - (long long)g.unary[(size_t)u * g.C + alpha] : 0LL);
        mine |= (l != alpha);
    }
    GC_DBG_ADD(0, 1)
    if (g.dbg && threadIdx.x == 0) g.dbg[15] = (long long)wall_clock64();
    if (!__syncthreads_or(mine)) return false;
    // pairwise terms (energy.h add_term2 / add_term1)
    for (int j = threadIdx.x; j < g.E; j += blockDim.x) {
        int p = g.edges[2 * j], q = g.edges[2 * j + 1];
        long long w = g.w[j];
        int lp = g.labels[p], lq = g.labels[q];
        int apq = g.edge_arc[2 * j], aqp = g.edge_arc[2 * j + 1];
        int cpq = 0, cqp = 0;
        bool ap = lp != alpha, aq = lq != alpha;
        const int32_t *V = g.smooth;
        if (ap && aq) {
            long long A = w * V[alpha * g.C + alpha], B = w * V[alpha * g.C + lq];
            long long Cc = w * V[lp * g.C + alpha], D = w * V[lp * g.C + lq];
            long long trp = D - A, trq = 0;
            B -= A;
            Cc -= D;
            if (B < 0) {
                trp -= B;
                trq += B;
                cqp = (int)(B + Cc);
            } else if (Cc < 0) {
                trp += Cc;
                trq -= Cc;
                cpq = (int)(B + Cc);
            } else {
                cpq = (int)B;
                cqp = (int)Cc;
            }
            if (trp) atomic_add_i64(&excess[p], trp);
            if (trq) atomic_add_i64(&excess[q], trq);
        } else if (ap) {
            long long d = w * V[lp * g.C + lq] - w * V[alpha * g.C + lq];
            if (d) atomic_add_i64(&excess[p], d);
        } else if (aq) {
            long long d = w * V[lp * g.C + lq] - w * V[lp * g.C + alpha];
            if (d) atomic_add_i64(&excess[q], d);
        }
        st(&cap[apq], cpq);
        st(&cap[aqp], cqp);
    }
    __syncthreads();

    GC_DBG_CLOCK(8)            // move set-up
    // maximum preflow
    for (int outer = 0;; ++outer) {
        GC_DBG_ADD(1, 1)
        if (outer > (1 << 20)) {
            if (threadIdx.x == 0) *g.status = 1;
            break;
        }
        gc_global_relabel<NPT>(g, t, cap, height, excess, alpha, flags, calls);
        int active = 0;
        for (int u = threadIdx.x; u < g.K; u += blockDim.x)
            if (ld(&excess[u]) > 0 && ld(&height[u]) < HMAX) active = 1;
        GC_DBG_CLOCK(9)        // global relabel
        if (!block_or(active, flags, calls)) break;
        for (int round = 0; round < GC_PUSH_ROUNDS; ++round) {
            GC_DBG_ADD(3, 1)
            int busy = 0;
            if (NPT > 0) {
#pragma unroll
                for (int s = 0; s < (NPT > 0 ? NPT : 1); ++s) {
                    const int u = threadIdx.x + s * blockDim.x;
                    if (u >= g.K) continue;
                    const long long e = ld(&excess[u]);
                    const int hu = ld(&height[u]);
                    if (e <= 0 || hu >= HMAX) continue;
                    busy = 1;
                    int best_h = 0x7fffffff, best_a = -1, best_v = 0, best_r = 0;
                    if (t.deg[s] > 0) {
                        int c[GC_ARCS], h[GC_ARCS];
#pragma unroll
                        for (int i = 0; i < GC_ARCS; ++i) {
                            c[i] = ld(&cap[t.a0[s] + min(i, t.deg[s] - 1)]);
                            h[i] = ld(&height[t.tr[s][i] & 0xffffu]);
                        }
#pragma unroll
                        for (int i = 0; i < GC_ARCS; ++i)
                            if (i < t.deg[s] && c[i] > 0 && h[i] < best_h) {       // (the first arc to the lowest neighbour)
                                best_h = h[i];
                                best_a = t.a0[s] + i;
                                best_v = (int)(t.tr[s][i] & 0xffffu);
                                best_r = (int)(t.tr[s][i] >> 16);
                            }
                        for (int a = t.a0[s] + GC_ARCS; a < t.a0[s] + t.deg[s]; ++a)
                            if (ld(&cap[a]) > 0) {
                                const int v = g.arc_to[a], h2 = ld(&height[v]);
                                if (h2 < best_h) {
                                    best_h = h2;
                                    best_a = a;
                                    best_v = v;
                                    best_r = g.arc_rev[a];
                                }
                            }
                    }
                    if (best_a < 0) {
                        st(&height[u], HMAX);
                    } else if (hu > best_h) {
                        const int c = ld(&cap[best_a]);
                        const int d = (e < (long long)c) ? (int)e : c;
                        atomicSub(&cap[best_a], d);
                        atomicAdd(&cap[best_r], d);
                        atomic_add_i64(&excess[u], -(long long)d);
                        atomic_add_i64(&excess[best_v], (long long)d);
                    } else {
                        st(&height[u], best_h + 1 < HMAX ? best_h + 1 : HMAX);
                    }
                }
            } else {
            for (int u = threadIdx.x; u < g.K; u += blockDim.x) {
                long long e = ld(&excess[u]);
                int hu = ld(&height[u]);
                if (e <= 0 || hu >= HMAX) continue;
                busy = 1;
                int best_h = 0x7fffffff, best_a = -1;
                const int a0 = g.arc_start[u], a1 = g.arc_start[u + 1];
                for (int base = a0; base < a1; base += GC_ARCS) {          // (batched like the relabelling pass)
                    int c[GC_ARCS], h[GC_ARCS];
#pragma unroll
                    for (int i = 0; i < GC_ARCS; ++i) {
                        const int a = min(base + i, a1 - 1);
                        c[i] = ld(&cap[a]);
                        h[i] = g.arc_to[a];
                    }
#pragma unroll
                    for (int i = 0; i < GC_ARCS; ++i) h[i] = ld(&height[h[i]]);
#pragma unroll
                    for (int i = 0; i < GC_ARCS; ++i)
                        if (base + i < a1 && c[i] > 0 && h[i] < best_h) {       // (the first arc to the lowest neighbour)
                            best_h = h[i];
                            best_a = base + i;
                        }
                }
                if (best_a < 0) {
                    st(&height[u], HMAX);
                } else if (hu > best_h) {
                    int c = ld(&cap[best_a]);
                    int d = (e < (long long)c) ? (int)e : c;
                    int v = g.arc_to[best_a];
                    atomicSub(&cap[best_a], d);
                    atomicAdd(&cap[g.arc_rev[best_a]], d);
                    atomic_add_i64(&excess[u], -(long long)d);
                    atomic_add_i64(&excess[v], (long long)d);
                } else {
                    st(&height[u], best_h + 1 < HMAX ? best_h + 1 : HMAX);
                }
            }
            }
            // (no active node in a whole round: nothing can change any more before the next global relabel)
            if (!block_or(busy, flags, calls)) break;
        }
        GC_DBG_CLOCK(10)       // push rounds
    }
    // (the loop above always ends on a fresh global relabel: height < HMAX <=> can reach the sink)
    for (int u = threadIdx.x; u < g.K; u += blockDim.x) {
        int l = g.labels[u];
        g.prop[u] = (l != alpha && ld(&height[u]) >= HMAX) ? alpha : l;
    }
    __syncthreads();
    long long after = gc_energy(g, g.prop, scratch);
    GC_DBG_CLOCK(11)           // cut + energy
    bool accept = after < *energy;
    __syncthreads();
    if (accept) {
        for (int u = threadIdx.x; u < g.K; u += blockDim.x) g.labels[u] = g.prop[u];
        if (threadIdx.x == 0) *energy = after;
    }
    __syncthreads();
    return accept;
}

// LVL: what lives in LDS -- 0 nothing (scratch in global memory), 1 the arrays the moves modify, 2 + labels, proposal, unary and
// smoothness costs, 3 + arc_start and arc_to, 4 + arc_rev.  A template parameter, not a run-time switch: with the placement known at
// compile time the accesses are LDS instructions; through pointers that may be either they are flat loads of twice the latency,
// and a relabelling level or a push round is nothing but a chain of such loads and a barrier.
// NPT > 0: at most NPT nodes per thread, the first GC_ARCS arcs of each in registers (GcTopo).
template <int LVL, int NPT>
__global__ void __launch_bounds__(GC_THREADS) k_alpha_expansion(GcDevice g)
{
    {
        const size_t zs = g.zs;
        ZSHIFT(g.E_dev, zs); ZSHIFT(g.K_dev, zs); ZSHIFT(g.edges, zs); ZSHIFT(g.w, zs); ZSHIFT(g.unary, zs); ZSHIFT(g.smooth, zs);
        ZSHIFT(g.arc_start, zs); ZSHIFT(g.arc_to, zs); ZSHIFT(g.arc_rev, zs); ZSHIFT(g.edge_arc, zs); ZSHIFT(g.labels, zs);
        ZSHIFT(g.prop, zs); ZSHIFT(g.energy_out, zs); ZSHIFT(g.g_cap, zs); ZSHIFT(g.g_height, zs); ZSHIFT(g.g_excess, zs);
        ZSHIFT(g.status, zs);
        if (g.K_dev) g.K = min(*g.K_dev, g.K);
    }
    if (g.E_dev && *g.E_dev > g.E) {
        // more edges than the tables hold (the caller sees the flag of k_gc_terms and comes back with larger ones): arcs beyond
        // the tables must not be followed -- a defined labelling, nothing else
        for (int u = threadIdx.x; u < g.K; u += blockDim.x) g.labels[u] = 0;
        if (threadIdx.x == 0) *g.energy_out = 0;
        return;
    }
    if (g.E_dev) g.E = min(*g.E_dev, g.E);
    extern __shared__ __align__(16) unsigned char dyn[];
    __shared__ long long scratch[GC_THREADS / 64];
    __shared__ long long energy;
    __shared__ int flags[3];
    __shared__ int table[GC_MAX_LABELS];
    __shared__ int queue_sizes[GC_MAX_LABELS + 2];

    // (read from the arrays in global memory: the LDS copies below are not complete before the next barrier)
    GcTopo<NPT> topo;
    if (NPT > 0) {
#pragma unroll
        for (int s = 0; s < (NPT > 0 ? NPT : 1); ++s) {
            const int u = threadIdx.x + s * blockDim.x;
            const bool in = u < g.K;
            topo.a0[s] = in ? g.arc_start[u] : 0;
            topo.deg[s] = in ? g.arc_start[u + 1] - topo.a0[s] : 0;
#pragma unroll
            for (int i = 0; i < GC_ARCS; ++i) {
                const int a = topo.a0[s] + min(i, topo.deg[s] - 1);
                topo.tr[s][i] = topo.deg[s] > 0 ? (unsigned)g.arc_to[a] | ((unsigned)g.arc_rev[a] << 16) : 0u;
            }
        }
    }
    long long *excess;
    int *cap, *height;
    int32_t *labels_out = g.labels;
    if (LVL >= 1) {
        // LDS: the arrays the moves modify, then -- as far as they fit -- what they only read.  A level
        // of the relabelling BFS or a push round is a chain of dependent reads per node (arc range -> neighbour -> its height):
        // ~2 us out of global memory, a fraction of that out of LDS.
        excess = reinterpret_cast<long long *>(dyn);
        cap = reinterpret_cast<int *>(dyn + (size_t)g.K * 8);
        height = cap + 2 * (size_t)g.e_cap;
        int *next = height + g.K;
        if (LVL >= 2) {
            int *lab = next, *prop = lab + g.K, *un = prop + g.K, *sm = un + (size_t)g.K * g.C;
            next = sm + g.C * g.C;
            for (int i = threadIdx.x; i < g.K * g.C; i += blockDim.x) un[i] = g.unary[i];
            for (int i = threadIdx.x; i < g.C * g.C; i += blockDim.x) sm[i] = g.smooth[i];
            g.labels = lab;
            g.prop = prop;
            g.unary = un;
            g.smooth = sm;
        }
        if (LVL >= 3) {
            int *as = next, *at = as + g.K + 1;
            next = at + 2 * (size_t)g.e_cap;
            for (int i = threadIdx.x; i <= g.K; i += blockDim.x) as[i] = g.arc_start[i];
            for (int i = threadIdx.x; i < 2 * g.E; i += blockDim.x) at[i] = g.arc_to[i];
            g.arc_start = as;
            g.arc_to = at;
            if (LVL >= 4) {
                int *ar = next;
                for (int i = threadIdx.x; i < 2 * g.E; i += blockDim.x) ar[i] = g.arc_rev[i];
                g.arc_rev = ar;
            }
        }
    } else {
        excess = g.g_excess;
        cap = g.g_cap;
        height = g.g_height;
    }
    unsigned calls = 0;
    if (threadIdx.x < 3) flags[threadIdx.x] = 0;
    for (int u = threadIdx.x; u < g.K; u += blockDim.x) g.labels[u] = 0;
    if (threadIdx.x < g.C) table[threadIdx.x] = threadIdx.x;
    __syncthreads();
    long long e0 = gc_energy(g, g.labels, scratch);
    if (threadIdx.x == 0) energy = e0;
    __syncthreads();

    if (g.n_iter == -1) {
        // GCoptimization::expansion(-1): adaptive cycles (see oracle orc_alpha_expansion_int)
        int nq = 1, next = 0;
        // label of the last move that was accepted: with a metric smoothness term (every move energy submodular, the max-flow
        // its exact optimum) expanding it again right away cannot lower the energy (the moves open to the new labelling are a
        // subset of those the accepted move was the optimum of), so that move is answered without a max-flow; with any other
        // matrix GCO truncates terms, a move is not exactly optimal, and every move is run (g.skip_repeat = 0)
        int last_accepted = -1;
        if (threadIdx.x == 0) queue_sizes[0] = g.C;
        __syncthreads();
        do {
            int queue_size = queue_sizes[nq - 1];
            int start = next;
            do {
                int alpha = table[next];
                bool ok = !(g.skip_repeat && alpha == last_accepted) && gc_expand<NPT>(g, topo, alpha, cap, height, excess, &energy, flags, calls, scratch);
                if (ok) last_accepted = alpha;
                if (!ok) {
                    --queue_size;
                    __syncthreads();
                    if (threadIdx.x == 0) {
                        int t = table[next];
                        table[next] = table[queue_size];
                        table[queue_size] = t;
                    }
                    __syncthreads();
                } else {
                    ++next;
                }
            } while (next < queue_size);
            int back = queue_sizes[nq - 1];
            __syncthreads();
            if (next == start) {
                next = back;
                nq--;
            } else if (queue_size < back / 2) {
                next = 0;
                if (threadIdx.x == 0) queue_sizes[nq] = queue_size;
                nq++;
            } else {
                next = 0;
            }
            __syncthreads();
        } while (nq > 0);
    } else {
        int last_accepted = -1;
        for (int cycle = 0; cycle < g.n_iter; ++cycle) {
            long long before = energy;
            __syncthreads();
            for (int l = 0; l < g.C; ++l)
                if (!(g.skip_repeat && table[l] == last_accepted) && gc_expand<NPT>(g, topo, table[l], cap, height, excess, &energy, flags, calls, scratch)) last_accepted = table[l];
            if (!(energy < before)) break;
        }
    }
    __syncthreads();
    if (g.labels != labels_out)
        for (int u = threadIdx.x; u < g.K; u += blockDim.x) labels_out[u] = g.labels[u];
    if (threadIdx.x == 0) *g.energy_out = energy;
}

// ---- graphs beyond one workgroup's reach: the same alpha-expansion by the WHOLE device ----------------------------------------
// A graph whose mutable arrays do not fit the LDS of one CU (K >~ 4 000 sites; the 298 116 supervoxels of BASELINE configs[4]) ran
// in ONE workgroup out of global memory: a level of the relabelling BFS or a push round was a scan of all K sites by 1 024 threads,
// 211 ms per volume.  k_alpha_expansion_grid is the same algorithm -- the same schedule of moves, the same move energies, the same
// lock-free push-relabel, the same cut convention, hence the same labelling -- on a cooperative launch: the sites are spread over
// all resident threads of the device, a workgroup barrier becomes a grid barrier, the workgroup-wide OR / sum go through three
// rotating words of a control block in global memory (rotation as block_or: one barrier per reduction).  Everything another
// workgroup may have written is read with agent-scope atomic loads (ld / st), as the single-workgroup kernel reads its global
// scratch.  The schedule variables live in registers of every thread and evolve identically (every decision is grid uniform).
//
// The grid barrier is the library's own (round 5; cooperative_groups' grid.sync() is a software barrier of ~26 us at 256
// workgroups on this runtime -- MI355X_MICROARCH.md price list, row barrier-cg -- and a move is a chain of hundreds of them).  Every
// word another workgroup reads is an agent-scope access (write-through stores, loads that bypass L1, atomics at the memory side), so
// the barrier needs no cache maintenance: every wave waits for its own stores (vmcnt(0)), the workgroup meets, ONE lane arrives.
// Round 6: ONE 64-bit word per barrier, and the OR of `any` rides on it -- a workgroup adds 1 + (its OR << 32) to the word of the
// epoch's parity with an atomic that returns nothing and polls the SAME word until its low half shows everybody (counts are
// monotonic; the word of the other parity is untouched until everybody has left this barrier, so all read the same final value); the
// high half -- how many workgroups have raised their OR so far -- against its value two epochs ago is the OR of this epoch.  One
// trip to memory after the last arrival, where round 5's barrier (a counter per group of workgroups, a top counter, release words,
// then the flag read back) took four or five: config 5's cut 24 -> see profiles/README_r06.md.
// Every wait is BOUNDED (2 s of the 100 MHz clock): a workgroup that gives up -- a block of the grid is not resident: the device is
// shared with another process, ADVICE r4 -- raises `poisoned`, which every waiting workgroup looks at between polls; all of them
// run the schedule down without waiting or changing anything, and the host cuts the graph again with the single workgroup.
struct GcGridCtl {
    int poisoned;                       // a barrier gave up: the result of this launch is void
    int pad0[31];
    unsigned long long word[2][16];     // one 128-byte line per barrier word
    long long sums[3];
    long long energy;
    int table[GC_MAX_LABELS];
    int queue_sizes[GC_MAX_LABELS + 2];
};
constexpr long long GC_GRID_WAIT_TICKS = 200000000LL;         // 2 s at 100 MHz

struct GcGrid {
    GcGridCtl *ctl;
    int tid, nth;
    unsigned sum_calls, epoch;
    unsigned raised[2];                 // (thread 0) the high half of each word when its last barrier completed
    bool dead;
    int *lds;                           // four LDS words of the workgroup: [0..2] its OR in turn, [3] the outcome of the wait
    __device__ GcGrid(GcGridCtl *c, int *lds_words)
        : ctl(c), tid(blockIdx.x * blockDim.x + threadIdx.x), nth(gridDim.x * blockDim.x), sum_calls(0), epoch(0), dead(false), lds(lds_words)
    {
        raised[0] = raised[1] = 0;
        if (threadIdx.x < 4) lds[threadIdx.x] = 0;
        __syncthreads();
    }
    // barrier + OR over the grid
    __device__ __forceinline__ bool any(int pred)
    {
        if (dead) return false;
        const unsigned e = epoch++;
        int *mine = lds + e % 3;                                           // (set and read by call e; cleared by thread 0 behind call e + 1's barrier)
        if (pred) *mine = 1;
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");        // every store / atomic of this wave has been performed
        __syncthreads();
        if (threadIdx.x == 0) {
            lds[(e + 2) % 3] = 0;
            unsigned long long *w = &ctl->word[e & 1][0];
            const unsigned target = (e / 2 + 1) * gridDim.x;          // (2^32 arrivals: beyond any cut)
            __hip_atomic_fetch_add(w, 1ULL + ((unsigned long long)(*mine != 0) << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            int outcome = -1;
            long long t0 = 0;
            for (unsigned spins = 1;; ++spins) {
                const unsigned long long v = __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if ((unsigned)v == target) {
                    outcome = (unsigned)(v >> 32) != raised[e & 1] ? 1 : 0;
                    raised[e & 1] = (unsigned)(v >> 32);
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
                if ((spins & 255u) == 0) {
                    if (__hip_atomic_load(&ctl->poisoned, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
                    const long long now = (long long)wall_clock64();
                    if (t0 == 0) t0 = now;
                    else if (now - t0 > GC_GRID_WAIT_TICKS) {
                        __hip_atomic_store(&ctl->poisoned, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        break;
                    }
                }
            }
            lds[3] = outcome;
        }
        __syncthreads();
        const int outcome = lds[3];          // (rewritten behind the first __syncthreads of the next barrier)
        if (outcome < 0) dead = true;
        return outcome > 0;
    }
    __device__ __forceinline__ void sync() { any(0); }
    // sum over the grid, one barrier: a wave adds its total with one atomic
    __device__ __forceinline__ long long sum(long long v)
    {
        long long *acc = ctl->sums + sum_calls % 3;
        v = wave_sum_i64(v);
        if ((threadIdx.x & 63) == 0 && v != 0 && !dead) atomic_add_i64(acc, v);
        sync();
        const long long r = ld(acc);
        if (tid == 0 && !dead) st(ctl->sums + (sum_calls + 2) % 3, 0LL);
        ++sum_calls;
        return r;
    }
};

__device__ __forceinline__ long long gc_grid_energy(const GcDevice &g, GcGrid &q, const int32_t *lab)
{
    long long e = 0;
    for (int i = q.tid; i < g.K; i += q.nth) e += g.unary[(size_t)i * g.C + ld(&lab[i])];
    for (int j = q.tid; j < g.E; j += q.nth) {
        const int a = g.edges[2 * j], b = g.edges[2 * j + 1];
        e += (long long)g.w[j] * g.smooth[ld(&lab[a]) * g.C + ld(&lab[b])];
    }
    return q.sum(e);
}

// Global relabelling = breadth-first distances to the sink in the residual graph.  Round 6: by the FRONTIER -- a node at `level` walks
// its arcs once and lifts the unreached neighbours that can push to it (residual capacity on the reverse arc) to level + 1 -- where
// rounds 4 / 5 had every unreached node walk all its arcs at every level, looking for a neighbour at `level`: at config 5 (298 116
// sites, ~14 arcs each, 67 levels per relabelling) that was ~30 walks per node instead of one, every word of them an agent-scope load
// that no cache on the way may serve -- 43 us per level, 20 of the cut's 24 ms.  Same levels, same heights (several frontier nodes may
// lift the same neighbour: to the same value), same end: a level that lifts nothing.
__device__ __forceinline__ void gc_grid_relabel(const GcDevice &g, GcGrid &q, int *cap, int *height, long long *excess, int alpha)
{
    const int HMAX = g.K + 2;
    for (int u = q.tid; u < g.K; u += q.nth)
        st(&height[u], (ld(&g.labels[u]) != alpha && ld(&excess[u]) < 0) ? 1 : HMAX);
    q.sync();
    for (int level = 1; level < HMAX; ++level) {
        if (g.dbg && q.tid == 0) g.dbg[2] += 1;
        int changed = 0;
        // (two nodes of a thread per turn: their heights are requested together -- a thread has one or two nodes at 256 workgroups)
        for (int v0 = q.tid; v0 < g.K; v0 += 2 * q.nth) {
            const int v1 = v0 + q.nth;
            const int h0 = ld(&height[v0]), h1 = v1 < g.K ? ld(&height[v1]) : -1;
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int v = t ? v1 : v0;
                if ((t ? h1 : h0) != level) continue;
                const int a0 = g.arc_start[v], a1 = g.arc_start[v + 1];
                for (int base = a0; base < a1; base += GC_ARCS) {
                    int to[GC_ARCS], c[GC_ARCS], h[GC_ARCS];
    #pragma unroll
                    for (int i = 0; i < GC_ARCS; ++i) {
                        const int a = min(base + i, a1 - 1);
                        to[i] = g.arc_to[a];
                        c[i] = g.arc_rev[a];
                    }
    #pragma unroll
                    for (int i = 0; i < GC_ARCS; ++i) {
                        c[i] = ld(&cap[c[i]]);                       // what the neighbour may still push along its arc to v
                        h[i] = ld(&height[to[i]]);
                    }
                    // (a neighbour that carries alpha has no capacity on any arc -- gc_grid_expand -- so it is never lifted)
    #pragma unroll
                    for (int i = 0; i < GC_ARCS; ++i)
                        if (base + i < a1 && c[i] > 0 && h[i] == HMAX) {
                            st(&height[to[i]], level + 1);
                            changed = 1;
                        }
                }
            }
        }
        if (!q.any(changed)) break;
    }
}

// one expansion move by the grid; returns (uniformly) whether the energy strictly decreased -- gc_expand, statement by statement
__device__ __forceinline__ bool gc_grid_expand(const GcDevice &g, GcGrid &q, int alpha, int *cap, int *height, long long *excess)
{
    const int HMAX = g.K + 2;
    int mine = 0;
    for (int u = q.tid; u < g.K; u += q.nth) {
        const int l = ld(&g.labels[u]);
        st(&excess[u], (l != alpha) ? (long long)g.unary[(size_t)u * g.C + l] - (long long)g.unary[(size_t)u * g.C + alpha] : 0LL);
        mine |= (l != alpha);
    }
    if (g.dbg && q.tid == 0) g.dbg[0] += 1;
    if (!q.any(mine)) return false;
    for (int j = q.tid; j < g.E; j += q.nth) {
        const int p = g.edges[2 * j], r = g.edges[2 * j + 1];
        const long long w = g.w[j];
        const int lp = ld(&g.labels[p]), lq = ld(&g.labels[r]);
        const int apq = g.edge_arc[2 * j], aqp = g.edge_arc[2 * j + 1];
        int cpq = 0, cqp = 0;
        const bool ap = lp != alpha, aq = lq != alpha;
        const int32_t *V = g.smooth;
        if (ap && aq) {
            long long A = w * V[alpha * g.C + alpha], B = w * V[alpha * g.C + lq];
            long long Cc = w * V[lp * g.C + alpha], D = w * V[lp * g.C + lq];
            long long trp = D - A, trq = 0;
            B -= A;
            Cc -= D;
            if (B < 0) {
                trp -= B;
                trq += B;
                cqp = (int)(B + Cc);
            } else if (Cc < 0) {
                trp += Cc;
                trq -= Cc;
                cpq = (int)(B + Cc);
            } else {
                cpq = (int)B;
                cqp = (int)Cc;
            }
            if (trp) atomic_add_i64(&excess[p], trp);
            if (trq) atomic_add_i64(&excess[r], trq);
        } else if (ap) {
            const long long d = w * V[lp * g.C + lq] - w * V[alpha * g.C + lq];
            if (d) atomic_add_i64(&excess[p], d);
        } else if (aq) {
            const long long d = w * V[lp * g.C + lq] - w * V[lp * g.C + alpha];
            if (d) atomic_add_i64(&excess[r], d);
        }
        st(&cap[apq], cpq);
        st(&cap[aqp], cqp);
    }
    q.sync();
    for (int outer = 0;; ++outer) {
        if (g.dbg && q.tid == 0) g.dbg[1] += 1;
        if (outer > (1 << 20)) {
            if (q.tid == 0) *g.status = 1;
            break;
        }
        gc_grid_relabel(g, q, cap, height, excess, alpha);
        int active = 0;
        for (int u = q.tid; u < g.K; u += q.nth)
            if (ld(&excess[u]) > 0 && ld(&height[u]) < HMAX) active = 1;
        if (!q.any(active)) break;
        for (int round = 0; round < GC_PUSH_ROUNDS; ++round) {
            if (g.dbg && q.tid == 0) g.dbg[3] += 1;
            int busy = 0;
            for (int u = q.tid; u < g.K; u += q.nth) {
                const long long e = ld(&excess[u]);
                const int hu = ld(&height[u]);
                if (e <= 0 || hu >= HMAX) continue;
                busy = 1;
                int best_h = 0x7fffffff, best_a = -1;
                const int a0 = g.arc_start[u], a1 = g.arc_start[u + 1];
                for (int base = a0; base < a1; base += GC_ARCS) {
                    int c[GC_ARCS], h[GC_ARCS];
#pragma unroll
                    for (int i = 0; i < GC_ARCS; ++i) {
                        const int a = min(base + i, a1 - 1);
                        c[i] = ld(&cap[a]);
                        h[i] = g.arc_to[a];
                    }
#pragma unroll
                    for (int i = 0; i < GC_ARCS; ++i) h[i] = ld(&height[h[i]]);
#pragma unroll
                    for (int i = 0; i < GC_ARCS; ++i)
                        if (base + i < a1 && c[i] > 0 && h[i] < best_h) {       // (the first arc to the lowest neighbour)
                            best_h = h[i];
                            best_a = base + i;
                        }
                }
                if (best_a < 0) {
                    st(&height[u], HMAX);
                } else if (hu > best_h) {
                    const int c = ld(&cap[best_a]);
                    const int d = (e < (long long)c) ? (int)e : c;
                    const int v = g.arc_to[best_a];
                    atomicSub(&cap[best_a], d);
                    atomicAdd(&cap[g.arc_rev[best_a]], d);
                    atomic_add_i64(&excess[u], -(long long)d);
                    atomic_add_i64(&excess[v], (long long)d);
                } else {
                    st(&height[u], best_h + 1 < HMAX ? best_h + 1 : HMAX);
                }
            }
            if (!q.any(busy)) break;
        }
    }
    // (the loop above always ends on a fresh global relabel: height < HMAX <=> can reach the sink)
    for (int u = q.tid; u < g.K; u += q.nth) {
        const int l = ld(&g.labels[u]);
        st(&g.prop[u], (l != alpha && ld(&height[u]) >= HMAX) ? alpha : l);
    }
    q.sync();
    const long long after = gc_grid_energy(g, q, g.prop);
    const bool accept = !q.dead && after < ld(&q.ctl->energy);
    q.sync();                   // (everybody has compared before the energy changes)
    if (accept) {
        for (int u = q.tid; u < g.K; u += q.nth) st(&g.labels[u], ld(&g.prop[u]));
        if (q.tid == 0) st(&q.ctl->energy, after);
    }
    q.sync();
    return accept;
}

// INVARIANT of this kernel: its grid barrier (GcGrid::sync) orders NOTHING but agent-scope accesses -- relaxed atomics
// plus s_waitcnt, no release / acquire fences, no cache maintenance.  Every word one workgroup writes and another reads between two
// barriers therefore MUST go through ld() / st() / the atomics of this file (agent scope: they bypass the non-coherent per-CU L1 and
// per-XCD L2 paths).  A plain load or store of shared data added here would be a silent race across XCDs.
__global__ void __launch_bounds__(GC_THREADS) k_alpha_expansion_grid(GcDevice g, GcGridCtl *ctl)
{
    if (g.K_dev) g.K = min(*g.K_dev, g.K);
    __shared__ int grid_words[4];
    if (g.test_absent && blockIdx.x == gridDim.x - 1 && gridDim.x > 1) return;
    GcGrid q(ctl, grid_words);
    if (g.E_dev && *g.E_dev > g.E) {            // (as k_alpha_expansion: more edges than the tables hold -> a defined labelling)
        for (int u = q.tid; u < g.K; u += q.nth) g.labels[u] = 0;
        if (q.tid == 0) *g.energy_out = 0;
        return;
    }
    if (g.E_dev) g.E = min(*g.E_dev, g.E);
    long long *excess = g.g_excess;
    int *cap = g.g_cap, *height = g.g_height;
    for (int u = q.tid; u < g.K; u += q.nth) st(&g.labels[u], 0);
    if (q.tid < g.C) st(&ctl->table[q.tid], q.tid);
    if (q.tid == 0) st(&ctl->queue_sizes[0], g.C);
    q.sync();
    const long long e0 = gc_grid_energy(g, q, g.labels);
    if (q.tid == 0) st(&ctl->energy, e0);
    q.sync();
    if (g.n_iter == -1) {
        // GCoptimization::expansion(-1): the adaptive cycles of k_alpha_expansion, the label table in the control block
        int nq = 1, next = 0, last_accepted = -1;
        do {
            int queue_size = ld(&ctl->queue_sizes[nq - 1]);
            const int start = next;
            do {
                const int alpha = ld(&ctl->table[next]);
                const bool ok = !(g.skip_repeat && alpha == last_accepted) && gc_grid_expand(g, q, alpha, cap, height, excess);
                if (ok) {
                    last_accepted = alpha;
                    ++next;
                } else {
                    --queue_size;
                    q.sync();
                    if (q.tid == 0) {
                        const int t = ld(&ctl->table[next]);
                        st(&ctl->table[next], ld(&ctl->table[queue_size]));
                        st(&ctl->table[queue_size], t);
                    }
                    q.sync();
                }
            } while (next < queue_size);
            const int back = ld(&ctl->queue_sizes[nq - 1]);
            q.sync();
            if (next == start) {
                next = back;
                nq--;
            } else if (queue_size < back / 2) {
                next = 0;
                if (q.tid == 0) st(&ctl->queue_sizes[nq], queue_size);
                nq++;
            } else {
                next = 0;
            }
            q.sync();
        } while (nq > 0);
    } else {
        int last_accepted = -1;
        for (int cycle = 0; cycle < g.n_iter; ++cycle) {
            const long long before = ld(&ctl->energy);
            q.sync();
            for (int l = 0; l < g.C; ++l)
                if (!(g.skip_repeat && l == last_accepted) && gc_grid_expand(g, q, l, cap, height, excess)) last_accepted = l;
            if (!(ld(&ctl->energy) < before)) break;
        }
    }
    q.sync();
    if (q.tid == 0 && !q.dead) *g.energy_out = ld(&ctl->energy);
}

// data costs only (GCO solveSpecialCases): independent argmin, first minimum wins
__global__ void k_unary_argmin(const int32_t *unary, int K, int C, int32_t *labels, long long *energy_out)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= K) return;
    int best = 0;
    for (int l = 1; l < C; ++l)
        if (unary[(size_t)i * C + l] < unary[(size_t)i * C + best]) best = l;
    labels[i] = best;
    atomic_add_i64(energy_out, unary[(size_t)i * C + best]);
}

static size_t gc_ctl_offset(int K, int E) { return ((size_t)K * 8 + ((size_t)K * 2 + (size_t)E * 2) * 4 + 63) & ~(size_t)63; }

size_t alpha_expansion_work_bytes(int K, int E)
{
    // g_excess[K] (8-byte aligned first) | prop[K] | g_height[K] | g_cap[2E] | the control block of the grid-wide kernel
    return gc_ctl_offset(K, E) + sizeof(GcGridCtl) + 64;
}

static std::atomic<long> g_grid_fallbacks{0};
long gc_grid_fallbacks() { return g_grid_fallbacks.load(); }

int launch_alpha_expansion(GcProblem p, const int32_t *arc_start, const int32_t *arc_to, const int32_t *arc_rev,
                           const int32_t *edge_arc, int n_iter, int32_t *labels_dev, long long *energy_dev,
                           int32_t *status_dev, void *work, hipStream_t st, ZBatch zb)
{
    if (p.C > GC_MAX_LABELS) {
        set_error("alpha_expansion: more than 64 labels are not supported by the single-workgroup kernel");
        return -1;
    }
    if (zb.nz > 1 && (!p.E_dev || !p.K_dev)) {
        set_error("alpha_expansion: a batch needs the site and edge counts on the device");
        return -1;
    }
    if (p.E == 0 && !p.E_dev) {
        HIP_TRY(hipMemsetAsync(energy_dev, 0, sizeof(long long), st));
        hipLaunchKernelGGL(k_unary_argmin, cdiv(p.K, 256), 256, 0, st, p.unary, p.K, p.C, labels_dev, energy_dev);
        HIP_TRY(hipGetLastError());
        return 0;
    }
    GcDevice g;
    g.K = p.K; g.C = p.C; g.E = p.E; g.n_iter = n_iter;
    g.E_dev = p.E_dev;
    g.edges = p.edges; g.w = p.w; g.unary = p.unary; g.smooth = p.smooth;
    g.arc_start = arc_start; g.arc_to = arc_to; g.arc_rev = arc_rev; g.edge_arc = edge_arc;
    g.labels = labels_dev;
    g.energy_out = energy_dev;
    g.status = status_dev;
    g.skip_repeat = p.metric;
    g.K_dev = p.K_dev;
    g.zs = zb.zs;
    g.test_absent = 0;
    g.dbg = nullptr;
    static const bool debug = getenv("IMSEGM_GC_DEBUG") != nullptr;
    static long long *dbg_buf = nullptr;
    if (debug && zb.nz == 1) {
        if (!dbg_buf) HIP_TRY(hipMalloc(&dbg_buf, 16 * sizeof(long long)));
        HIP_TRY(hipStreamSynchronize(st));
        long long h[16];
        HIP_TRY(hipMemcpy(h, dbg_buf, sizeof(h), hipMemcpyDeviceToHost));
        if (h[0] > 0 && h[0] < (1 << 20))
            fprintf(stderr, "[alpha expansion] previous call: %lld moves, %lld relabels, %lld BFS levels, %lld push rounds; us: set-up %.1f relabel %.1f "
                            "push %.1f cut+energy %.1f\n", h[0], h[1], h[2], h[3], h[8] / 100.0, h[9] / 100.0, h[10] / 100.0, h[11] / 100.0);
        HIP_TRY(hipMemset(dbg_buf, 0, 16 * sizeof(long long)));
        g.dbg = dbg_buf;
    }
    unsigned char *wb = (unsigned char *)work;
    g.g_excess = (long long *)wb;
    g.prop = (int32_t *)(wb + (size_t)p.K * 8);
    g.g_height = g.prop + p.K;
    g.g_cap = g.g_height + p.K;
    const size_t lds_max = 150 * 1024;
    size_t lds_need = (size_t)p.K * 8 + ((size_t)2 * p.E + p.K) * 4;
    g.use_lds = lds_need <= lds_max;
    g.e_cap = p.E;
    g.lds_lab = g.lds_topo = 0;
    int level = 0;                            // (nested: each level on top of the one before, as far as 150 KB reach)
    if (g.use_lds) {
        level = 1;
        const size_t lab = ((size_t)2 * p.K + (size_t)p.K * p.C + (size_t)p.C * p.C) * 4, topo1 = ((size_t)p.K + 1 + (size_t)2 * p.E) * 4, topo2 = (size_t)2 * p.E * 4;
        if (lds_need + lab <= lds_max) {
            level = 2;
            g.lds_lab = 1;
            lds_need += lab;
            if (lds_need + topo1 <= lds_max) {
                level = 3;
                g.lds_topo = 1;
                lds_need += topo1;
                if (lds_need + topo2 <= lds_max) {
                    level = 4;
                    g.lds_topo = 2;
                    lds_need += topo2;
                }
            }
        }
    }
    const int level_cap = knobs().gc_lds_level;      // (tests: every placement)
    if (level > level_cap) {
        level = std::max(0, level_cap);
        g.use_lds = level >= 1; g.lds_lab = level >= 2; g.lds_topo = level >= 4 ? 2 : level >= 3 ? 1 : 0;
        lds_need = (size_t)p.K * 8 + ((size_t)2 * p.E + p.K) * 4;
        if (level >= 2) lds_need += ((size_t)2 * p.K + (size_t)p.K * p.C + (size_t)p.C * p.C) * 4;
        if (level >= 3) lds_need += ((size_t)p.K + 1 + (size_t)2 * p.E) * 4;
    }
    // a graph that does not fit the LDS of one CU: the whole device works on it (k_alpha_expansion_grid) -- one workgroup of 1 024
    // threads per CU at most (a grid barrier is one atomic per workgroup on one word), a site or two per thread
    if (level == 0 && zb.nz == 1 && p.K >= (knobs().gc_grid_min_sites > 0 ? knobs().gc_grid_min_sites : 8192) && !knobs().gc_one_workgroup) {
        int dev = 0;
        HIP_TRY(hipGetDevice(&dev));
        // One grid-wide cut at a time PER DEVICE of this process (two cooperative grids dispatched from two streams of one device
        // at once could each get a part of its CUs and wait at their first barrier for the rest; cuts on different devices
        // cannot starve each other and are not serialised).  The lock also covers the first-use query of the device below.  Another
        // PROCESS on the same GPU is not covered by any lock: there a workgroup that never becomes resident makes the barrier give
        // up after two seconds (GcGridCtl::poisoned) and the graph is cut by the single workgroup below -- ranks should not share
        // a GPU, IMSEGM_GC_ONE_WORKGROUP=1 avoids the wait where they must.
        static std::mutex one_at_a_time[IMSEGM_MAX_DEVICES];
        static int cus[IMSEGM_MAX_DEVICES] = { 0 };
        // a launch that gave up (the device is shared: two seconds lost) is remembered per device: the next cuts go straight to
        // the single workgroup and the grid is tried again after a while, instead of paying the wait on every cut
        static int sit_out[IMSEGM_MAX_DEVICES] = { 0 };
        const bool known = dev >= 0 && dev < IMSEGM_MAX_DEVICES;
        std::unique_lock<std::mutex> guard;
        if (known) guard = std::unique_lock<std::mutex>(one_at_a_time[dev]);
        if (known && cus[dev] == 0) {
            hipDeviceProp_t prop;
            HIP_TRY(hipGetDeviceProperties(&prop, dev));
            int per_cu = 0;
            HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void *)k_alpha_expansion_grid, GC_THREADS, 0));
            cus[dev] = prop.cooperativeLaunch && per_cu >= 1 ? prop.multiProcessorCount : -1;
        }
        int n_cu = known ? cus[dev] : -1;
        if (known && sit_out[dev] > 0) {
            --sit_out[dev];
            n_cu = -1;
        }
        if (n_cu > 0) {
            GcGridCtl *ctl = reinterpret_cast<GcGridCtl *>(wb + gc_ctl_offset(p.K, p.E));
            HIP_TRY(hipMemsetAsync(ctl, 0, sizeof(GcGridCtl), st));
            int blocks = std::min(n_cu, cdiv(p.K, GC_THREADS));
            if (knobs().gc_grid_blocks > 0) blocks = std::min(blocks, knobs().gc_grid_blocks);
            g.test_absent = knobs().gc_grid_test_absent ? 1 : 0;
            void *args[] = { &g, &ctl };
            // The launch is followed to its end here (the callers synchronise a few launches later anyway; a cut of this size takes
            // milliseconds).  A cooperative launch: the runtime checks the grid against what the device can hold at once.
            HIP_TRY(hipLaunchCooperativeKernel((const void *)k_alpha_expansion_grid, dim3(blocks), dim3(GC_THREADS), args, 0, st));
            int poisoned = 0;
            HIP_TRY(hipMemcpyAsync(&poisoned, &ctl->poisoned, sizeof(int), hipMemcpyDeviceToHost, st));
            HIP_TRY(hipStreamSynchronize(st));
            if (!poisoned) return 0;
            g_grid_fallbacks.fetch_add(1);
            if (known && !knobs().gc_grid_test_absent) sit_out[dev] = 16;
            // (not all workgroups of the grid were resident: the single workgroup below starts from scratch)
        }
        if (guard.owns_lock()) guard.unlock();
    }
    size_t dyn = g.use_lds ? lds_need : 0;
    // one thread per node up to 1024; a small graph runs with fewer waves (the moves are chains of workgroup barriers)
    int threads = std::min(GC_THREADS, std::max(256, ((p.K + 63) / 64) * 64));
    if (knobs().gc_threads) threads = std::min(GC_THREADS, std::max(64, knobs().gc_threads & ~63));      // (experiments)
    // arcs in registers: one node per thread (two would spill at 1024 threads)
    const bool cached = level >= 2 && p.K <= threads && 2 * (long)p.E <= 0xffff && !knobs().gc_no_topo_regs;
#define GC_LAUNCH(L, N)                                                                                                           \
    {                                                                                                                             \
        if (dyn > 48 * 1024) /* the opt-in is per device and cheap: set it on every launch that needs it */                       \
            HIP_TRY(hipFuncSetAttribute((const void *)k_alpha_expansion<L, N>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024)); \
        hipLaunchKernelGGL((k_alpha_expansion<L, N>), dim3(1, 1, zb.nz), threads, dyn, st, g);                                    \
    }
    switch (level + (cached ? 10 : 0)) {
    case 0: GC_LAUNCH(0, 0) break;
    case 1: GC_LAUNCH(1, 0) break;
    case 2: GC_LAUNCH(2, 0) break;
    case 3: GC_LAUNCH(3, 0) break;
    case 4: GC_LAUNCH(4, 0) break;
    case 12: GC_LAUNCH(2, 1) break;
    case 13: GC_LAUNCH(3, 1) break;
    default: GC_LAUNCH(4, 1) break;
    }
#undef GC_LAUNCH
    HIP_TRY(hipGetLastError());
    return 0;
}

}  // namespace imsegm
