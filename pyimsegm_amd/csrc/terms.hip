// terms.hip -- everything between the descriptors and the graph cut, on the device: feature table, class
// probabilities of a Gaussian mixture, unary / edge terms, their pyGCO integer form and the CSR arc structure of the
// superpixel graph.  With these the whole chain  label map -> statistics -> graph -> terms -> alpha-expansion ->
// gathers  is enqueued on one stream without a host round trip (api.hip imsegm_image2d_segment).
//
// Replaces, in /root/reference/imsegm/graph_cuts.py:
//   model.predict_proba(features)  for sklearn Pipeline([StandardScaler,] GaussianMixture(covariance_type='full'))
//       (:73-163 estim_class_model builds exactly this; pipelines.py:96,232 call it)
//   compute_unary_cost :523-540, compute_edge_model :383-439, compute_spatial_dist :303-336,
//   compute_edge_weights :616-657 (edge types '', const, spatial, model[_l1|_l2|_lT], features), edge_cost :722
// and the float -> integer conversion of gco-wrapper's pygco.cut_general_graph (down_weight_factor, truncation).
// Arithmetic: fp64, one rounding per operation (-ffp-contract=off), formula order of numpy / scikit-learn; sums run in
// a fixed tree order (numpy uses pairwise summation, BLAS its own order): class probabilities agree with scikit-learn
// to ~1e-13, edge weights to ~1e-15 relative (tests: 1e-9 / 1e-12) -- the integer energies are identical unless a
// scaled cost lies within that distance of an integer.
#include "slic.h"

namespace imsegm {

constexpr int TM_THREADS = 1024;
constexpr int TERMS_WIDE_FROM = 16384;     // supervoxels from which the terms are computed by the whole device (launch_gc_terms)

__device__ __forceinline__ double block_reduce_f64(double v, double *scratch, bool is_max)
{
    // fixed order: lanes of a wave by xor butterfly, then the 16 wave results in index order
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const double o = __shfl_xor(v, off, 64);
        v = is_max ? fmax(v, o) : v + o;
    }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = v;
    __syncthreads();
    double t = scratch[0];
    for (int i = 1; i < TM_THREADS / 64; ++i) t = is_max ? fmax(t, scratch[i]) : t + scratch[i];
    __syncthreads();
    return t;
}

// ---- feature table [K][F] from the statistics of stats.hip: columns mean | std | energy (descriptors.py:787-863 order),
// np.nan_to_num and the -0 -> +0 of descriptors.py:1265 applied
__global__ void __launch_bounds__(256)
k_features_assemble(const double *__restrict__ mean, const double *__restrict__ energy, const double *__restrict__ var, int K,
                    int mask, int F, double *__restrict__ out, int col0, size_t zs)
{
    ZSHIFT(mean, zs); ZSHIFT(energy, zs); ZSHIFT(var, zs); ZSHIFT(out, zs);
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= K * 3) return;
    const int k = i / 3, ch = i - 3 * k;
    int col = col0 + ch;
    auto put = [&](double v) {
        if (v != v) v = 0.0;                                          // nan_to_num
        else if (v > DBL_MAX) v = DBL_MAX;
        else if (v < -DBL_MAX) v = -DBL_MAX;
        if (v == 0.0) v = 0.0;                                        // -0 -> +0
        out[(size_t)k * F + col] = v;
        col += 3;
    };
    if (mask & 1) put(mean[i]);
    if (mask & 2) put(sqrt(var[i]));
    if (mask & 4) put(energy[i]);
}

// ---- symmetric adjacency: the pixel pass sets bit (row b, column a) for a < b; mirror it so that row v lists ALL
// neighbours of v in ascending order -- the order in which the host CSR of round 1 held the arcs of v (edges sorted
// by (b, a): first the edges whose larger end is v, then those whose smaller end is v)
__global__ void __launch_bounds__(256)
k_adj_symmetrize(uint32_t *bitmap, const int *__restrict__ Kp, int words, size_t zs)
{
    ZSHIFT(bitmap, zs); ZSHIFT(Kp, zs);
    const int K = *Kp;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    for (int b = wave; b < K; b += (gridDim.x * blockDim.x) >> 6) {
        for (int w = lane; w <= (b >> 5) && w < words; w += 64) {
            uint32_t bits = bitmap[(size_t)b * words + w];
            if (w == (b >> 5)) bits &= (1u << (b & 31)) - 1u;         // columns below b only
            while (bits) {
                const int a = w * 32 + __ffs(bits) - 1;
                bits &= bits - 1;
                atomicOr(bitmap + (size_t)a * words + (b >> 5), 1u << (b & 31));
            }
        }
    }
}

// per row: exclusive popcount prefix per word, degree, number of lower neighbours
__global__ void __launch_bounds__(256)
k_adj_rowprefix(const uint32_t *__restrict__ bitmap, const int *__restrict__ Kp, int words, int32_t *__restrict__ wordprefix,
                int32_t *__restrict__ deg, int32_t *__restrict__ deg_low, size_t zs)
{
    ZSHIFT(bitmap, zs); ZSHIFT(Kp, zs); ZSHIFT(wordprefix, zs); ZSHIFT(deg, zs); ZSHIFT(deg_low, zs);
    const int K = *Kp;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    for (int v = wave; v < K; v += (gridDim.x * blockDim.x) >> 6) {
        int carry = 0, low = 0;
        for (int w0 = 0; w0 < words; w0 += 64) {
            const int w = w0 + lane;
            const uint32_t bits = w < words ? bitmap[(size_t)v * words + w] : 0u;
            const int c = __popc(bits);
            int incl = c;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const int t = __shfl_up(incl, off, 64);
                if (lane >= off) incl += t;
            }
            if (w < words) wordprefix[(size_t)v * words + w] = carry + incl - c;
            int lowc = 0;
            if (w < (v >> 5)) lowc = c;
            else if (w == (v >> 5)) lowc = __popc(bits & ((1u << (v & 31)) - 1u));
            low += wave_sum_i32(lowc);
            carry += __shfl(incl, 63, 64);
        }
        if (lane == 0) {
            deg[v] = carry;
            deg_low[v] = low;
        }
    }
}

// exclusive scans of deg_low (-> first edge index of row v) and deg (-> arc_start) by one workgroup
__global__ void __launch_bounds__(256)
k_adj_scan(const int *__restrict__ Kp, const int32_t *__restrict__ deg, const int32_t *__restrict__ deg_low, int32_t *arc_start,
           int32_t *edge_start, int32_t *n_edges, size_t zs)
{
    ZSHIFT(Kp, zs); ZSHIFT(deg, zs); ZSHIFT(deg_low, zs); ZSHIFT(arc_start, zs); ZSHIFT(edge_start, zs); ZSHIFT(n_edges, zs);
    __shared__ int wsum[2][4];
    __shared__ int carry[2];
    const int K = *Kp;
    if (threadIdx.x < 2) carry[threadIdx.x] = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // (four consecutive entries per lane and turn: a turn is a trip to memory and three barriers whatever it carries -- 1 165 turns of
    // 256 entries for the 298 116 supervoxels of config 5 were 0.94 ms)
    constexpr int PER = 4;
    for (int base = 0; base < K; base += 256 * PER) {
        const int i = base + threadIdx.x * PER;
        int v0[PER], v1[PER];
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            v0[j] = i + j < K ? deg[i + j] : 0;
            v1[j] = i + j < K ? deg_low[i + j] : 0;
        }
        int t0 = 0, t1 = 0;
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            t0 += v0[j];
            t1 += v1[j];
        }
        int i0 = t0, i1 = t1;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int u0 = __shfl_up(i0, off, 64), u1 = __shfl_up(i1, off, 64);
            if (lane >= off) {
                i0 += u0;
                i1 += u1;
            }
        }
        if (lane == 63) {
            wsum[0][wave] = i0;
            wsum[1][wave] = i1;
        }
        __syncthreads();
        int p0 = carry[0], p1 = carry[1];
        for (int w = 0; w < wave; ++w) {
            p0 += wsum[0][w];
            p1 += wsum[1][w];
        }
        int e0 = p0 + i0 - t0, e1 = p1 + i1 - t1;             // exclusive prefix of the lane's first entry
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            if (i + j < K) {
                arc_start[i + j] = e0;
                edge_start[i + j] = e1;
            }
            e0 += v0[j];
            e1 += v1[j];
        }
        __syncthreads();
        if (threadIdx.x == 255) {
            carry[0] = p0 + i0;
            carry[1] = p1 + i1;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        arc_start[K] = carry[0];
        *n_edges = carry[1];
    }
}

// arcs, reverse arcs, edge list (a < b, ordered by (b, a)) and the edge -> arc table
__global__ void __launch_bounds__(256)
k_adj_emit(const uint32_t *__restrict__ bitmap, const int *__restrict__ Kp, int words, const int32_t *__restrict__ wordprefix,
           const int32_t *__restrict__ arc_start, const int32_t *__restrict__ edge_start, int edge_capacity,
           int32_t *__restrict__ edges, int32_t *__restrict__ arc_to, int32_t *__restrict__ arc_rev, int32_t *__restrict__ edge_arc,
           size_t zs)
{
    ZSHIFT(bitmap, zs); ZSHIFT(Kp, zs); ZSHIFT(wordprefix, zs); ZSHIFT(arc_start, zs); ZSHIFT(edge_start, zs); ZSHIFT(edges, zs);
    ZSHIFT(arc_to, zs); ZSHIFT(arc_rev, zs); ZSHIFT(edge_arc, zs);
    const int K = *Kp;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    for (int v = wave; v < K; v += (gridDim.x * blockDim.x) >> 6) {
        const int a0 = arc_start[v], e0 = edge_start[v];
        for (int w = lane; w < words; w += 64) {
            uint32_t bits = bitmap[(size_t)v * words + w];
            int t = wordprefix[(size_t)v * words + w];
            while (bits) {
                const int u = w * 32 + __ffs(bits) - 1;
                bits &= bits - 1;
                const int pos = a0 + t;
                const int rev = arc_start[u] + wordprefix[(size_t)u * words + (v >> 5)] +
                                __popc(bitmap[(size_t)u * words + (v >> 5)] & ((1u << (v & 31)) - 1u));
                if (pos < 2 * edge_capacity) {
                    arc_to[pos] = u;
                    arc_rev[pos] = rev;
                }
                if (u < v) {                                    // lower neighbours come first: t is the rank inside the row
                    const int j = e0 + t;
                    if (j < edge_capacity) {
                        edges[2 * j] = u;
                        edges[2 * j + 1] = v;
                        edge_arc[2 * j] = rev;                  // arc u -> v
                        edge_arc[2 * j + 1] = pos;              // arc v -> u
                    }
                }
                ++t;
            }
        }
    }
}

// ---- the same arcs out of the symmetric neighbour table of a label volume (volume.hip k_vol_adjacency_runs<1>; round 6: the
// 3 * 10^5 supervoxels of BASELINE configs[4] made the bitmap 11 GB, its word prefixes another 11 GB, and the three kernels above
// scan them -- the table is K x 64 slots = 76 MB).  One wave per row, one slot per lane (cap <= 64):
//   k_tab_sort_rows  ranks the entries of a row among themselves and writes them back in ascending order (free slots behind);
//                    degree and number of smaller neighbours fall out of two votes;
//   k_adj_scan       (above) turns them into arc_start / edge_start / the edge count;
//   k_tab_emit       lane i of row v holds arc v -> u_i: its reverse arc is the position of v in the (sorted) row of u -- a binary
//                    search over at most 64 slots --, and u_i < v makes it edge (u_i, v) number edge_start[v] + i.
// Same arcs in the same order as the bitmap path gives (rows ascending, neighbours ascending).
__global__ void __launch_bounds__(256)
k_tab_sort_rows(int32_t *table, const int *__restrict__ Kp, int cap, int32_t *__restrict__ deg, int32_t *__restrict__ deg_low,
                const int *__restrict__ overflow)
{
    const int K = *Kp;
    const int v = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (v >= K) return;
    if (*overflow) {
        // a row was too narrow for its label's neighbours: the table is not the graph.  The caller reports it after its
        // synchronisation (IMSEGM_E_FUSED_PATH); until then everything downstream works on a graph without edges.
        if (lane == 0) {
            deg[v] = 0;
            deg_low[v] = 0;
        }
        return;
    }
    int32_t *row = table + (size_t)v * cap;
    const int e = lane < cap ? row[lane] : -1;
    const bool valid = e >= 0;
    int rank = 0;
    for (int j = 0; j < cap; ++j) {
        const int o = __builtin_amdgcn_readlane(e, j);
        rank += (o >= 0 && o < e) ? 1 : 0;
    }
    const int cnt = __popcll(__ballot(valid)), low = __popcll(__ballot(valid && e < v));
    // (every lane has read its slot before any lane writes: the rank loop consumed the loads)
    if (valid) row[rank] = e;
    else if (lane < cap) {
        // the free slots move behind the entries: lane = position among the invalid lanes
        const unsigned long long inv = __ballot(!valid && lane < cap);
        row[cnt + __popcll(inv & ((1ULL << lane) - 1ULL))] = -1;
    }
    if (lane == 0) {
        deg[v] = cnt;
        deg_low[v] = low;
    }
}

__global__ void __launch_bounds__(256)
k_tab_emit(const int32_t *__restrict__ table, const int *__restrict__ Kp, int cap, const int32_t *__restrict__ arc_start,
           const int32_t *__restrict__ edge_start, int edge_capacity, int32_t *__restrict__ edges, int32_t *__restrict__ arc_to,
           int32_t *__restrict__ arc_rev, int32_t *__restrict__ edge_arc)
{
    const int K = *Kp;
    const int v = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (v >= K) return;
    if (lane >= arc_start[v + 1] - arc_start[v]) return;          // (the sorted row holds its deg entries in front)
    const int u = table[(size_t)v * cap + lane];
    const int pos = arc_start[v] + lane;
    const int32_t *ru = table + (size_t)u * cap;
    int lo = 0, hi = arc_start[u + 1] - arc_start[u];              // position of v in the sorted row of u
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (ru[mid] < v) lo = mid + 1;
        else hi = mid;
    }
    const int rev = arc_start[u] + lo;
    if (pos < 2 * edge_capacity) {
        arc_to[pos] = u;
        arc_rev[pos] = rev;
    }
    if (u < v) {                                                  // the smaller neighbours come first: lane = rank inside the row
        const int j = edge_start[v] + lane;
        if (j < edge_capacity) {
            edges[2 * j] = u;
            edges[2 * j + 1] = v;
            edge_arc[2 * j] = rev;                               // arc u -> v
            edge_arc[2 * j + 1] = pos;                           // arc v -> u
        }
    }
}

int launch_graph_csr_table(int32_t *table, const int *K_dev, int K_cap, int cap, const int *overflow, int32_t *deg, int32_t *deg_low,
                           int32_t *arc_start, int32_t *edge_start, int32_t *n_edges_dev, int edge_capacity, int32_t *edges,
                           int32_t *arc_to, int32_t *arc_rev, int32_t *edge_arc, hipStream_t st)
{
    if (cap < 1 || cap > 64) {
        set_error("graph from the neighbour table: rows of 1 .. 64 slots");
        return -1;
    }
    const dim3 grid(cdiv((long)K_cap * 64, 256));
    hipLaunchKernelGGL(k_tab_sort_rows, grid, 256, 0, st, table, K_dev, cap, deg, deg_low, overflow);
    hipLaunchKernelGGL(k_adj_scan, dim3(1, 1, 1), 256, 0, st, K_dev, deg, deg_low, arc_start, edge_start, n_edges_dev, (size_t)0);
    hipLaunchKernelGGL(k_tab_emit, grid, 256, 0, st, table, K_dev, cap, arc_start, edge_start, edge_capacity, edges, arc_to, arc_rev,
                       edge_arc);
    HIP_TRY(hipGetLastError());
    return 0;
}

// ---- class probabilities + graph-cut terms: ONE workgroup (K ~ 2e3 rows, E ~ 6e3 edges: a few microseconds; the
// phases need grid-wide reductions of a handful of scalars, which a single workgroup gets from __syncthreads)

// predict_proba of Pipeline([StandardScaler,] GaussianMixture('full')): StandardScaler.transform,
// GaussianMixture._estimate_weighted_log_prob, scipy logsumexp, exp.  One wave per superpixel: lane j forms
// y_j = sum_f x_f P[c][f][j] - mu_proj[c][j] (sequential in f, as a plain dot product), lane 0 adds the squares in index order.
// several images per launch (ZBatch): every pointer of the argument block moves to image blockIdx.z (the parameter block with the
// class model is uploaded once per image, so the model pointers move too)
__device__ __forceinline__ void zshift_terms(TermsArgs &a)
{
    const size_t zs = a.zs;
    ZSHIFT(a.Kp, zs); ZSHIFT(a.Ep, zs); ZSHIFT(a.features, zs); ZSHIFT(a.scaler_mean, zs); ZSHIFT(a.scaler_scale, zs);
    ZSHIFT(a.prec_chol, zs); ZSHIFT(a.mu_proj, zs); ZSHIFT(a.log_det, zs); ZSHIFT(a.log_w, zs); ZSHIFT(a.proba, zs);
    ZSHIFT(a.edges, zs); ZSHIFT(a.centres, zs); ZSHIFT(a.edge_dist, zs); ZSHIFT(a.edge_len, zs); ZSHIFT(a.unary, zs);
    ZSHIFT(a.weights, zs); ZSHIFT(a.pairwise, zs); ZSHIFT(a.unary_i, zs); ZSHIFT(a.weights_i, zs); ZSHIFT(a.status, zs);
    ZSHIFT(a.scalars, zs); ZSHIFT(a.fstd, zs);
}

// predict_proba of the mixture; lane l keeps the features (and the projected coordinates) l, l + 64, ... (NF of them: F <= 64 NF)
// of the GMM_SPW superpixels its wave works on (4 beyond 64 features, else 1) -- a row of the precision factor is read once for all
// of them (with one superpixel
// per wave the 2 000 waves of a 2048^2 image read the 778 KB of a 180-feature model 2 000 times: 0.62 ms, bound by the L2).
// Sums run in ascending feature order, per superpixel exactly as the one-superpixel kernel of the earlier rounds formed them.
template <int NF, int GMM_SPW>
__global__ void __launch_bounds__(256) k_gmm_proba(TermsArgs a)
{
    zshift_terms(a);
    const int K = *a.Kp, C = a.C, F = a.F;
    const int lane = threadIdx.x & 63;
    const int k0 = ((blockIdx.x * blockDim.x + threadIdx.x) >> 6) * GMM_SPW;
    if (k0 >= K) return;
    double xv[GMM_SPW][NF];
#pragma unroll
    for (int s = 0; s < GMM_SPW; ++s) {
        const int k = min(k0 + s, K - 1);                // (the last wave repeats the last superpixel; only k < K is written)
#pragma unroll
        for (int i = 0; i < NF; ++i) {
            const int f = lane + 64 * i;
            double v = 0.0;
            if (f < F) {
                v = a.features[(size_t)k * F + f];
                if (a.scaler_mean) v = v - a.scaler_mean[f];
                if (a.scaler_scale) v = v / a.scaler_scale[f];
            }
            xv[s][i] = v;
        }
    }
    double mywl[GMM_SPW], amax[GMM_SPW];                 // lane c keeps the weighted log probability of class c
#pragma unroll
    for (int s = 0; s < GMM_SPW; ++s) mywl[s] = amax[s] = -INFINITY;
    for (int c = 0; c < C; ++c) {
        const double *P = a.prec_chol + (size_t)c * F * F;
        double y[GMM_SPW][NF];
#pragma unroll
        for (int s = 0; s < GMM_SPW; ++s)
#pragma unroll
            for (int j = 0; j < NF; ++j) y[s][j] = 0.0;
#pragma unroll
        for (int i = 0; i < NF; ++i) {
            const int count = min(64, F - 64 * i);       // (uniform)
            for (int l = 0; l < count; ++l) {
                const double *row = P + (size_t)(64 * i + l) * F;
                double pr[NF];
#pragma unroll
                for (int j = 0; j < NF; ++j) pr[j] = lane + 64 * j < F ? row[lane + 64 * j] : 0.0;
#pragma unroll
                for (int s = 0; s < GMM_SPW; ++s) {
                    const double xf = __shfl(xv[s][i], l, 64);
#pragma unroll
                    for (int j = 0; j < NF; ++j)
                        if (lane + 64 * j < F) y[s][j] += xf * pr[j];
                }
            }
        }
#pragma unroll
        for (int s = 0; s < GMM_SPW; ++s) {
            double lp = 0.0;
#pragma unroll
            for (int j = 0; j < NF; ++j) {
                double d = 0.0;
                if (lane + 64 * j < F) d = y[s][j] - a.mu_proj[c * F + lane + 64 * j];
                const double y2 = d * d;
                const int count = min(64, F - 64 * j);
                for (int l = 0; l < count; ++l) lp += __shfl(y2, l, 64);
            }
            const double lg = -0.5 * (a.const_term + lp) + a.log_det[c];
            const double w = lg + a.log_w[c];
            if (lane == c) mywl[s] = w;
            amax[s] = fmax(amax[s], w);
        }
    }
#pragma unroll
    for (int s = 0; s < GMM_SPW; ++s) {
        double top = amax[s];
        if (!(fabs(top) <= DBL_MAX)) top = 0.0;          // scipy.special.logsumexp: non-finite maximum -> 0
        const double ex = exp(mywl[s] - top);            // lanes >= C: exp(-inf) = 0
        double sum = 0.0;
        for (int c = 0; c < C; ++c) sum += __shfl(ex, c, 64);
        const double lse = log(sum) + top;
        if (lane < C && k0 + s < K) a.proba[(size_t)(k0 + s) * C + lane] = exp(mywl[s] - lse);
    }
}

__global__ void __launch_bounds__(TM_THREADS) k_gc_terms(TermsArgs a)
{
    zshift_terms(a);
    __shared__ double scratch[TM_THREADS / 64];
    const int K = *a.Kp, C = a.C, F = a.F;
    int E = *a.Ep;
    if (E > a.edge_capacity) {
        if (threadIdx.x == 0) atomicOr(a.status, 2);
        E = a.edge_capacity;
    }
    // (1. predict_proba ran before this kernel: k_gmm_proba, one wave per superpixel)
    // 2. unary cost |-log(clip(p, 0.01, 0.99))| and its maximum
    double umax = 0.0;
    for (int i = threadIdx.x; i < K * C; i += TM_THREADS) {
        double p = a.proba[i];
        if (p < 0.01) p = 0.01;
        if (p > 1 - 0.01) p = 1 - 0.01;
        const double u = fabs(-log(p));
        a.unary[i] = u;
        umax = fmax(umax, u);
    }
    umax = block_reduce_f64(umax, scratch, true);
    // 3. edge type 'features': StandardScaler().fit_transform(features) -- per-column mean and population std
    if (a.edge_type == 5) {
        for (int f = 0; f < F; ++f) {
            double s = 0.0;
            for (int k = threadIdx.x; k < K; k += TM_THREADS) s += a.features[(size_t)k * F + f];
            const double mean = block_reduce_f64(s, scratch, false) / (double)K;
            double q = 0.0;
            for (int k = threadIdx.x; k < K; k += TM_THREADS) {
                const double d = a.features[(size_t)k * F + f] - mean;
                q += d * d;
            }
            double sd = sqrt(block_reduce_f64(q, scratch, false) / (double)K);
            if (sd == 0.0) sd = 1.0;                              // sklearn _handle_zeros_in_scale
            if (threadIdx.x == 0) {
                a.fstd[f] = mean;
                a.fstd[F + f] = sd;
            }
        }
        __syncthreads();
    }
    // 4. per edge: distance of the end points in the chosen space + Euclidean distance of the centres
    double sum_len = 0.0, sum_dist = 0.0;
    for (int j = threadIdx.x; j < E; j += TM_THREADS) {
        const int p = a.edges[2 * j], q = a.edges[2 * j + 1];
        double len = 0.0;
        for (int d = 0; d < a.ndim; ++d) {
            double cp = a.centres[(size_t)p * a.ndim + d], cq = a.centres[(size_t)q * a.ndim + d];
            const double t = cp - cq;
            len += t * t;
        }
        len = sqrt(len);
        double dist = 0.0;
        if (a.edge_type >= 2 && a.edge_type <= 4) {
            for (int c = 0; c < C; ++c) {
                const double t = a.proba[(size_t)p * C + c] - a.proba[(size_t)q * C + c];
                if (a.edge_type == 2) dist = fmax(dist, t * t);           // lT: max squared difference
                else if (a.edge_type == 3) dist += fabs(t);               // l1
                else dist += t * t;                                       // l2 (root below)
            }
            if (a.edge_type == 4) dist = sqrt(dist);
        } else if (a.edge_type == 5) {
            for (int f = 0; f < F; ++f) {
                const double xp = (a.features[(size_t)p * F + f] - a.fstd[f]) / a.fstd[F + f];
                const double xq = (a.features[(size_t)q * F + f] - a.fstd[f]) / a.fstd[F + f];
                const double t = xp - xq;
                dist += t * t;
            }
            dist = sqrt(dist);
        }
        a.edge_len[j] = len;
        a.edge_dist[j] = dist;
        sum_len += len;
        sum_dist += dist;
    }
    const double mean_len = block_reduce_f64(sum_len, scratch, false) / (double)E;
    const double mean_dist = block_reduce_f64(sum_dist, scratch, false) / (double)E;
    double q = 0.0;
    for (int j = threadIdx.x; j < E; j += TM_THREADS) {
        const double d = a.edge_dist[j] - mean_dist;
        q += d * d;
    }
    const double std_dist = sqrt(block_reduce_f64(q, scratch, false) / (double)E);
    // 5. weights: exp(-dist / (2 std^2)) | 1, divided by the relative centre distance, clipped, times edge_cost
    const double denom = 2 * (std_dist * std_dist);
    double wmax = 0.0;
    for (int j = threadIdx.x; j < E; j += TM_THREADS) {
        double w = 1.0;
        if (a.edge_type >= 2) w = exp(-a.edge_dist[j] / denom);
        if (a.spatial_norm) w = w / (a.edge_len[j] / mean_len);
        if (w < 1. / 1e3) w = 1. / 1e3;
        if (w > 1e3) w = 1e3;
        w = w * a.edge_cost;
        a.weights[j] = w;
        wmax = fmax(wmax, fabs(w));
    }
    wmax = block_reduce_f64(wmax, scratch, true);
    // 6. pygco.cut_general_graph: down_weight_factor, integer energies by truncation
    const double dwf = ((E > 0 && wmax * a.pairwise_max > umax) ? wmax * a.pairwise_max : umax) + 1e-10;
    for (int i = threadIdx.x; i < K * C; i += TM_THREADS) a.unary_i[i] = (int32_t)((a.unary[i] / dwf) * 100000);
    int bad = 0;
    for (int j = threadIdx.x; j < E; j += TM_THREADS) {
        const int32_t wi = (int32_t)((a.weights[j] / dwf) * 1000);
        a.weights_i[j] = wi;
        if ((long long)abs(wi) * a.smooth_max > 10000000LL) bad = 1;     // GCO_MAX_ENERGYTERM
    }
    if (bad) atomicOr(a.status, 1);
    if (threadIdx.x == 0) {
        a.scalars[0] = mean_len; a.scalars[1] = mean_dist; a.scalars[2] = std_dist;
        a.scalars[3] = umax; a.scalars[4] = wmax; a.scalars[5] = dwf;
    }
}

// ---- the same terms for the graph of a volume (round 6) ----------------------------------------------------------------------
// k_gc_terms is ONE workgroup: right for the 2 000 superpixels of an image, 4.8 ms for the 298 116 supervoxels and 2 * 10^6 edges of
// config 5 -- 2 000 turns of a loop whose every turn waits for its gathers.  Here the element-wise steps (2, 4, 5, 6) run on the whole
// device, and the sums keep THE ORDER of k_gc_terms: "thread" t of its 1 024 adds the elements t, t + 1 024, ... in ascending order
// (k_terms_partial: sixteen waves, each streaming its lanes' elements), and the 1 024 partial sums meet in block_reduce_f64 as they
// do there (k_terms_reduce) -- the same additions in the same order, hence the same bits in the means, the deviation, the weights
// and the integers (tests hold the two paths against each other).  Maxima do not depend on an order: atomics on the bit patterns of
// the non-negative values.
__device__ __forceinline__ int terms_edges(const TermsArgs &a) { return min(*a.Ep, a.edge_capacity); }

__device__ __forceinline__ void atomic_max_nonneg_f64(unsigned long long *bits, double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_xor(v, off, 64));
    if ((threadIdx.x & 63) == 0 && v > 0.0) atomicMax(bits, (unsigned long long)__double_as_longlong(v));
}

__global__ void __launch_bounds__(256) k_terms_elem(TermsArgs a, unsigned long long *maxbits)
{
    const int K = *a.Kp, C = a.C;
    const int E = terms_edges(a);
    const int tid = blockIdx.x * 256 + threadIdx.x, nth = gridDim.x * 256;
    if (tid == 0 && *a.Ep > a.edge_capacity) atomicOr(a.status, 2);
    double umax = 0.0;
    for (int i = tid; i < K * C; i += nth) {
        double p = a.proba[i];
        if (p < 0.01) p = 0.01;
        if (p > 1 - 0.01) p = 1 - 0.01;
        const double u = fabs(-log(p));
        a.unary[i] = u;
        umax = fmax(umax, u);
    }
    atomic_max_nonneg_f64(maxbits, umax);
    for (int j = tid; j < E; j += nth) {
        const int p = a.edges[2 * j], q = a.edges[2 * j + 1];
        double len = 0.0;
        for (int d = 0; d < a.ndim; ++d) {
            double cp = a.centres[(size_t)p * a.ndim + d], cq = a.centres[(size_t)q * a.ndim + d];
            const double t = cp - cq;
            len += t * t;
        }
        len = sqrt(len);
        double dist = 0.0;
        if (a.edge_type >= 2 && a.edge_type <= 4) {
            for (int c = 0; c < C; ++c) {
                const double t = a.proba[(size_t)p * C + c] - a.proba[(size_t)q * C + c];
                if (a.edge_type == 2) dist = fmax(dist, t * t);           // lT: max squared difference
                else if (a.edge_type == 3) dist += fabs(t);               // l1
                else dist += t * t;                                       // l2 (root below)
            }
            if (a.edge_type == 4) dist = sqrt(dist);
        }
        a.edge_len[j] = len;
        a.edge_dist[j] = dist;
    }
}

// MODE 0: the sums of the edge lengths and distances; MODE 1: of the squared deviations of the distances from their mean
template <int MODE> __global__ void __launch_bounds__(64) k_terms_partial(TermsArgs a, double *partials)
{
    const int E = terms_edges(a);
    const int t = blockIdx.x * 64 + threadIdx.x;              // the thread of k_gc_terms whose share this lane adds up
    if (MODE == 0) {
        double sum_len = 0.0, sum_dist = 0.0;
#pragma unroll 8
        for (int j = t; j < E; j += TM_THREADS) {
            sum_len += a.edge_len[j];
            sum_dist += a.edge_dist[j];
        }
        partials[t] = sum_len;
        partials[TM_THREADS + t] = sum_dist;
    } else {
        const double mean_dist = a.scalars[1];
        double q = 0.0;
#pragma unroll 8
        for (int j = t; j < E; j += TM_THREADS) {
            const double d = a.edge_dist[j] - mean_dist;
            q += d * d;
        }
        partials[t] = q;
    }
}

template <int MODE> __global__ void __launch_bounds__(TM_THREADS) k_terms_reduce(TermsArgs a, const double *partials)
{
    __shared__ double scratch[TM_THREADS / 64];
    const int E = terms_edges(a);
    if (MODE == 0) {
        const double mean_len = block_reduce_f64(partials[threadIdx.x], scratch, false) / (double)E;
        const double mean_dist = block_reduce_f64(partials[TM_THREADS + threadIdx.x], scratch, false) / (double)E;
        if (threadIdx.x == 0) {
            a.scalars[0] = mean_len;
            a.scalars[1] = mean_dist;
        }
    } else {
        const double std_dist = sqrt(block_reduce_f64(partials[threadIdx.x], scratch, false) / (double)E);
        if (threadIdx.x == 0) a.scalars[2] = std_dist;
    }
}

__global__ void __launch_bounds__(256) k_terms_weights(TermsArgs a, unsigned long long *maxbits)
{
    const int E = terms_edges(a);
    const int tid = blockIdx.x * 256 + threadIdx.x, nth = gridDim.x * 256;
    const double mean_len = a.scalars[0], std_dist = a.scalars[2];
    const double denom = 2 * (std_dist * std_dist);
    double wmax = 0.0;
    for (int j = tid; j < E; j += nth) {
        double w = 1.0;
        if (a.edge_type >= 2) w = exp(-a.edge_dist[j] / denom);
        if (a.spatial_norm) w = w / (a.edge_len[j] / mean_len);
        if (w < 1. / 1e3) w = 1. / 1e3;
        if (w > 1e3) w = 1e3;
        w = w * a.edge_cost;
        a.weights[j] = w;
        wmax = fmax(wmax, fabs(w));
    }
    atomic_max_nonneg_f64(maxbits + 1, wmax);
}

__global__ void __launch_bounds__(256) k_terms_integers(TermsArgs a, const unsigned long long *maxbits)
{
    const int K = *a.Kp, C = a.C;
    const int E = terms_edges(a);
    const int tid = blockIdx.x * 256 + threadIdx.x, nth = gridDim.x * 256;
    const double umax = __longlong_as_double((long long)maxbits[0]), wmax = __longlong_as_double((long long)maxbits[1]);
    const double dwf = ((E > 0 && wmax * a.pairwise_max > umax) ? wmax * a.pairwise_max : umax) + 1e-10;
    for (int i = tid; i < K * C; i += nth) a.unary_i[i] = (int32_t)((a.unary[i] / dwf) * 100000);
    int bad = 0;
    for (int j = tid; j < E; j += nth) {
        const int32_t wi = (int32_t)((a.weights[j] / dwf) * 1000);
        a.weights_i[j] = wi;
        if ((long long)abs(wi) * a.smooth_max > 10000000LL) bad = 1;     // GCO_MAX_ENERGYTERM
    }
    if (bad) atomicOr(a.status, 1);
    if (tid == 0) {
        a.scalars[3] = umax; a.scalars[4] = wmax; a.scalars[5] = dwf;
    }
}

// gc_regul <= 0: argmin of the unary cost (graph_cuts.py:729-731), first minimum wins as np.argmin
__global__ void __launch_bounds__(256)
k_unary_argmin_f64(const double *__restrict__ unary, const int *__restrict__ Kp, int C, int32_t *__restrict__ labels, size_t zs)
{
    ZSHIFT(unary, zs); ZSHIFT(Kp, zs); ZSHIFT(labels, zs);
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= *Kp) return;
    int best = 0;
    for (int c = 1; c < C; ++c)
        if (unary[(size_t)k * C + c] < unary[(size_t)k * C + best]) best = c;
    labels[k] = best;
}

// LUT of the final gather: classes_[graph_labels] (pipelines.py:238) or the graph labels themselves
__global__ void __launch_bounds__(256)
k_label_lut(const int32_t *__restrict__ graph_labels, const int *__restrict__ Kp, const int32_t *__restrict__ classes,
            int32_t *__restrict__ lut, size_t zs)
{
    ZSHIFT(graph_labels, zs); ZSHIFT(Kp, zs); ZSHIFT(classes, zs); ZSHIFT(lut, zs);
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= *Kp) return;
    const int l = graph_labels[k];
    lut[k] = classes ? classes[l] : l;
}

int launch_features_assemble(const double *mean, const double *energy, const double *var, int K, int mask, double *out,
                             hipStream_t st, int row_stride, int col0, ZBatch zb)
{
    // (row_stride: columns of the table the block is written into -- several blocks side by side; 0: the block is the table)
    const int F = row_stride > 0 ? row_stride : 3 * (((mask & 1) != 0) + ((mask & 2) != 0) + ((mask & 4) != 0));
    hipLaunchKernelGGL(k_features_assemble, dim3(cdiv((long)K * 3, 256), 1, zb.nz), 256, 0, st, mean, energy, var, K, mask, F, out, col0, zb.zs);
    HIP_TRY(hipGetLastError());
    return 0;
}

int launch_graph_csr(uint32_t *bitmap, const int *K_dev, int K_cap, int words, int32_t *wordprefix, int32_t *deg, int32_t *deg_low,
                     int32_t *arc_start, int32_t *edge_start, int32_t *n_edges_dev, int edge_capacity, int32_t *edges,
                     int32_t *arc_to, int32_t *arc_rev, int32_t *edge_arc, hipStream_t st, ZBatch zb)
{
    const dim3 grid(std::min(cdiv((long)K_cap * 64, 256), 4096), 1, zb.nz);
    hipLaunchKernelGGL(k_adj_symmetrize, grid, 256, 0, st, bitmap, K_dev, words, zb.zs);
    hipLaunchKernelGGL(k_adj_rowprefix, grid, 256, 0, st, bitmap, K_dev, words, wordprefix, deg, deg_low, zb.zs);
    hipLaunchKernelGGL(k_adj_scan, dim3(1, 1, zb.nz), 256, 0, st, K_dev, deg, deg_low, arc_start, edge_start, n_edges_dev, zb.zs);
    hipLaunchKernelGGL(k_adj_emit, grid, 256, 0, st, bitmap, K_dev, words, wordprefix, arc_start, edge_start, edge_capacity,
                       edges, arc_to, arc_rev, edge_arc, zb.zs);
    HIP_TRY(hipGetLastError());
    return 0;
}

int launch_gc_terms(const TermsArgs &a, hipStream_t st, int nz)
{
    if (a.F > 256 || a.C > 16) {
        set_error("device class model: at most 256 features and 16 classes");
        return -1;
    }
    if (a.gmm) {
        // (few features: a wave per superpixel -- the model is a few hundred bytes and more waves hide more latency: 13 against
        // 22 us for 2 000 x 9; many features: four superpixels per wave share the rows of the precision factor)
        const int spw = a.F <= 64 ? 1 : 4;
        const dim3 grid(cdiv((long)cdiv(a.K_cap, spw) * 64, 256), 1, nz);
        if (a.F <= 64) hipLaunchKernelGGL((k_gmm_proba<1, 1>), grid, 256, 0, st, a);
        else if (a.F <= 128) hipLaunchKernelGGL((k_gmm_proba<2, 4>), grid, 256, 0, st, a);
        else if (a.F <= 192) hipLaunchKernelGGL((k_gmm_proba<3, 4>), grid, 256, 0, st, a);
        else hipLaunchKernelGGL((k_gmm_proba<4, 4>), grid, 256, 0, st, a);
    }
    if (nz == 1 && a.zs == 0 && a.edge_type != 5 && a.K_cap >= TERMS_WIDE_FROM && a.edge_capacity >= 2 * TM_THREADS && !knobs().terms_one_workgroup) {
        // the graph of a volume: the element-wise steps on the whole device, the sums in k_gc_terms' order (see k_terms_elem)
        unsigned long long *maxbits = reinterpret_cast<unsigned long long *>(a.scalars + 6);
        double *partials = a.weights;                     // (2 x 1 024 words of an array nobody reads before k_terms_weights fills it)
        const int wide = std::min(4096, cdiv(std::max(a.K_cap * a.C, a.edge_capacity), 256));
        HIP_TRY(hipMemsetAsync(maxbits, 0, 2 * sizeof(unsigned long long), st));
        hipLaunchKernelGGL(k_terms_elem, wide, 256, 0, st, a, maxbits);
        hipLaunchKernelGGL(k_terms_partial<0>, TM_THREADS / 64, 64, 0, st, a, partials);
        hipLaunchKernelGGL(k_terms_reduce<0>, 1, TM_THREADS, 0, st, a, (const double *)partials);
        hipLaunchKernelGGL(k_terms_partial<1>, TM_THREADS / 64, 64, 0, st, a, partials);
        hipLaunchKernelGGL(k_terms_reduce<1>, 1, TM_THREADS, 0, st, a, (const double *)partials);
        hipLaunchKernelGGL(k_terms_weights, wide, 256, 0, st, a, maxbits);
        hipLaunchKernelGGL(k_terms_integers, wide, 256, 0, st, a, (const unsigned long long *)maxbits);
        HIP_TRY(hipGetLastError());
        return 0;
    }
    hipLaunchKernelGGL(k_gc_terms, dim3(1, 1, nz), TM_THREADS, 0, st, a);
    HIP_TRY(hipGetLastError());
    return 0;
}

int launch_unary_argmin(const double *unary, const int *K_dev, int K_cap, int C, int32_t *labels, hipStream_t st, ZBatch zb)
{
    hipLaunchKernelGGL(k_unary_argmin_f64, dim3(cdiv(K_cap, 256), 1, zb.nz), 256, 0, st, unary, K_dev, C, labels, zb.zs);
    HIP_TRY(hipGetLastError());
    return 0;
}

int launch_label_lut(const int32_t *graph_labels, const int *K_dev, int K_cap, const int32_t *classes, int32_t *lut, hipStream_t st,
                     ZBatch zb)
{
    hipLaunchKernelGGL(k_label_lut, dim3(cdiv(K_cap, 256), 1, zb.nz), 256, 0, st, graph_labels, K_dev, classes, lut, zb.zs);
    HIP_TRY(hipGetLastError());
    return 0;
}

}  // namespace imsegm
