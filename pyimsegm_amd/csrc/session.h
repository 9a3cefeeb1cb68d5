// session.h -- the host-side objects behind the C ABI (include/imsegm_hip.h): context (device + stream + profiler), the
// device-resident state of one image / volume, and the small helpers api.hip and batch.hip share.
#pragma once
#include "../../include/imsegm_hip.h"
#include "slic.h"

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace imsegm {

// growable device buffer
struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    int ensure(size_t bytes)
    {
        if (bytes <= cap) return 0;
        if (p) HIP_TRY(hipFree(p));
        p = nullptr;
        cap = 0;
        size_t want = bytes + bytes / 8 + 256;
        HIP_TRY(hipMalloc(&p, want));
        cap = want;
        return 0;
    }
    void release()
    {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
    template <typename T> T *as() { return reinterpret_cast<T *>(p); }
};

enum { PG_ASSIGN = 0, PG_SLIC = 1, PG_CONN = 2, PG_STATS = 3, PG_GRAPH = 4, PG_GC = 5, PG_GATHER = 6, PG_PRE = 7, PG_TERMS = 8, PG_TEX = 9, PG_COUNT = 10 };

struct Span {
    int group;
    hipEvent_t a, b;
};

}  // namespace imsegm

using namespace imsegm;

struct imsegm_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool profile = false;
    std::vector<Span> spans;
    std::vector<hipEvent_t> pool;
    double acc_ms[PG_COUNT] = { 0 };
    int acc_n[PG_COUNT] = { 0 };
    DevBuf gc_buf;   // scratch of imsegm_cut_general_graph
    DevBuf aux_buf;  // small second scratch (border histogram of imsegm_assume_bg_on_boundary)
    void *pinned = nullptr;          // page-locked staging for the small host <-> device transfers
    size_t pinned_cap = 0;
    hipEvent_t pinned_ev = nullptr;   // recorded after an H2D out of `pinned` that nobody waits for
    bool pinned_busy = false;
    void mark_stage_in_flight()
    {
        if (!pinned_ev) (void)hipEventCreateWithFlags(&pinned_ev, hipEventDisableTiming);
        (void)hipEventRecord(pinned_ev, stream);
        pinned_busy = true;
    }
    void *stage(size_t bytes)
    {
        if (pinned_busy) {
            (void)hipEventSynchronize(pinned_ev);
            pinned_busy = false;
        }
        if (bytes > pinned_cap) {
            if (pinned) (void)hipHostFree(pinned);
            pinned = nullptr;
            pinned_cap = 0;
            size_t want = bytes + bytes / 2 + 4096;
            if (hipHostMalloc(&pinned, want, hipHostMallocDefault) != hipSuccess) return nullptr;
            pinned_cap = want;
        }
        return pinned;
    }

    hipEvent_t get_event()
    {
        if (!pool.empty()) {
            hipEvent_t e = pool.back();
            pool.pop_back();
            return e;
        }
        hipEvent_t e;
        (void)hipEventCreate(&e);
        return e;
    }
    int begin(int group)
    {
        if (!profile) return -1;
        Span s;
        s.group = group;
        s.a = get_event();
        s.b = get_event();
        (void)hipEventRecord(s.a, stream);
        spans.push_back(s);
        return (int)spans.size() - 1;
    }
    void end(int id)
    {
        if (id >= 0) (void)hipEventRecord(spans[id].b, stream);
    }
    void pair(int group, hipEvent_t *a, hipEvent_t *b)       // events filled in by a kernel launch, not recorded here
    {
        Span s;
        s.group = group;
        s.a = get_event();
        s.b = get_event();
        spans.push_back(s);
        *a = s.a;
        *b = s.b;
    }
    void collect()
    {
        if (spans.empty()) return;
        (void)hipStreamSynchronize(stream);
        for (auto &s : spans) {
            float ms = 0;
            if (hipEventElapsedTime(&ms, s.a, s.b) == hipSuccess) {
                acc_ms[s.group] += ms;
                acc_n[s.group] += 1;
            }
            pool.push_back(s.a);
            pool.push_back(s.b);
        }
        spans.clear();
    }
};

// where the graph of a session's label map lives inside its `gseg` buffer (api.hip graph_plan: the layout follows from the number
// of labels, the edge capacity and the form of the adjacency store -- K x K bitmap, or 64 neighbour slots per label for the label
// volumes whose bitmap would not be small)
struct GraphPlan {
    int K = 0, Ecap = 0, cap = 0, words = 0;
    bool table = false;
    size_t o_head = 0, o_edges = 0, o_as = 0, o_at = 0, o_ar = 0, o_ea = 0, o_deg = 0, o_dlow = 0, o_es = 0, o_cent = 0, o_present = 0,
           o_store = 0, o_cacc = 0, o_wp = 0, bytes = 0;
};

struct imsegm_image2d {
    imsegm_ctx *ctx = nullptr;
    int D = 1, H = 0, W = 0;      // D > 1: gray volume session (imsegm_volume_*)
    size_t n = 0;
    int dtype = -1;
    int n_labels = 0;
    bool have_labels = false;
    bool labels_connected = false;              // the label map is what the connectivity pass wrote: every label > 0 is ONE 6-connected set
    bool tex_ready = false;
    bool is_volume = false;
    double vol_off = 0.0, vol_scale = 1.0;      // intensity seen by the volume SLIC = (v + off) * scale
    DevBuf img, labA, labB, nearest, labels, conn_i32, conn_u8, small, cent, tiles, feat, graph, gather_lut, gather_out_i, gather_out_f,
        tex_planes, tex_resp, tex_small, vol_cent, annot, hist, featK, seg, gseg, narrow;
    GraphPlan gplan;                            // the graph imsegm_image2d_graph_prepare has enqueued into `gseg` ...
    bool graph_ready = false;                   // ... for the current label map (any call that changes the labels clears this)
    int *slic_fail_host = nullptr;              // page-locked word the centroid update inside the assignment kernel raises when it hands the image back
    int feat_mask = 0, feat_F = 0;              // layout of the resident feature table (imsegm_image2d_features_color)
    int place_F = 0, place_col = 0;             // imsegm_image2d_features_place: where the next descriptor call puts its columns
    // the ten SLIC sweeps (30 kernel launches + a memset) as one captured HIP graph, re-used while every launch parameter
    // (sizes, pointers, weights: `slic_key`, the bytes of a SlicGraphKey) stays the same -- a recycled session replays it
    hipGraphExec_t slic_exec = nullptr;
    std::vector<unsigned char> slic_key;
};

// entry points are specific to colour images (D == 1) or gray volumes (created by imsegm_volume_create)
inline int wrong_kind(const imsegm_image2d *im, bool want_volume)
{
    if (im && im->is_volume != want_volume) {
        set_error(want_volume ? "this call needs a volume session" : "this call needs a 2-D colour image session");
        return 1;
    }
    return 0;
}

extern std::atomic<bool> g_runtime_started;      // a HIP call has been made through this library (imsegm_init is too late); api.hip

inline int bind(imsegm_ctx *ctx)
{
    if (!ctx) {
        set_error("null context");
        return -1;
    }
    HIP_TRY(hipSetDevice(ctx->device));
    return 0;
}

// the session's page of reduction words: zeroed once where it is allocated (launch_minmax leaves its words at zero, slic.hip)
inline int ensure_small(imsegm_image2d *im)
{
    if (im->small.cap >= 4096) return 0;
    if (im->small.ensure(4096)) return -1;
    HIP_TRY(hipMemsetAsync(im->small.p, 0, 4096, im->ctx->stream));
    return 0;
}

// skimage.util.regular_grid (util/_regular_grid.py, 0.18) for a 3-D shape
struct GridAxis {
    long start, step;
    bool all;
};
inline void regular_grid3(const long shape[3], long n_points, GridAxis out[3])
{
    int order[3] = { 0, 1, 2 };
    std::stable_sort(order, order + 3, [&](int a, int b) { return shape[a] < shape[b]; });
    double sorted_dims[3] = { (double)shape[order[0]], (double)shape[order[1]], (double)shape[order[2]] };
    double space = sorted_dims[0] * sorted_dims[1] * sorted_dims[2];
    if (space <= (double)n_points) {
        for (int i = 0; i < 3; ++i) out[i] = { 0, 1, true };
        return;
    }
    double steps[3];
    for (int i = 0; i < 3; ++i) steps[i] = pow(space / (double)n_points, 1.0 / 3);
    bool any_small = false;
    for (int i = 0; i < 3; ++i) any_small |= sorted_dims[i] < steps[i];
    if (any_small) {
        for (int dim = 0; dim < 3; ++dim) {
            steps[dim] = sorted_dims[dim];
            double sp = 1.0;
            for (int j = dim + 1; j < 3; ++j) sp *= sorted_dims[j];
            if (dim < 2) {
                double s = pow(sp / (double)n_points, 1.0 / (3 - dim - 1));
                for (int j = dim + 1; j < 3; ++j) steps[j] = s;
            }
            bool ok = true;
            for (int j = 0; j < 3; ++j) ok &= sorted_dims[j] >= steps[j];
            if (ok) break;
        }
    }
    for (int i = 0; i < 3; ++i) {
        long start = (long)floor(steps[i] / 2.0);
        long step = (long)nearbyint(steps[i]);
        out[order[i]] = { start, step, false };
    }
}

inline int fill_taps(Taps &t, const double *w, int r)
{
    t.r = -1;
    for (int i = 0; i < 17; ++i) t.w[i] = 0;
    if (r < 0 || !w) return 0;
    if (r > 16) {
        set_error("gaussian kernel radius > 16 is not supported");
        return -1;
    }
    t.r = r;
    for (int i = 0; i <= r; ++i) t.w[i] = w[i];
    return 0;
}

inline bool is_pinned(const void *p)
{
    hipPointerAttribute_t attr;
    if (hipPointerGetAttributes(&attr, p) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    return attr.type == hipMemoryTypeHost;
}


// helpers shared by the files of the C ABI (api.hip was split by stage in round 6): defined in api_volume.hip, api_image2d.hip,
// api_fused.hip
extern "C" {
// segmented statistics of `src` on the session's label map (mean, energy, variance: any may be null); planar: [C][H][W] planes
int stats_run(imsegm_image2d *im, const void *src, int dtype, double maxabs, int planar, int prescale, double mul, double div,
              double *mean_out, double *energy_out, double *var_out, long plane_stride = -1);
// where a descriptor call puts its columns of the resident feature table (imsegm_image2d_features_place)
int take_placement(imsegm_image2d *im, int own_F, bool to_host, int *table_F, int *col0);
}

// what of the 2-D SLIC state follows from the sizes (api_image2d.hip)
struct SlicGeometry {
    int K;             // centroids of the regular grid
    size_t n_tiles;    // 64 x 32 candidate tiles
};
int slic_geometry(int H, int W, int n_segments, double compactness, int minmax_normalize, int max_candidates, int slic_zero,
                  imsegm::SlicState &s, SlicGeometry &geo);
void slic_place_state(imsegm::SlicState &s, const SlicGeometry &geo, unsigned char *cent, unsigned char *tiles, const double *premax,
                      int *fail_host);
inline size_t slic_cent_bytes(int K) { return (size_t)K * (5 * 8 + 16 + 9 * 8 + 16 + 4) + 256 + imsegm::SLIC_DRIFT_SLOTS * sizeof(int); }
inline size_t slic_tiles_bytes(size_t n_tiles, size_t n)
{
    return n_tiles * (imsegm::SLIC_MAXC * (sizeof(imsegm::Cand) + sizeof(imsegm::Rec32) + 2 * sizeof(int)) + sizeof(imsegm::TileInfo) + sizeof(int)) +
           n * 4 + 1024;
}
// the pieces of the connectivity scratch (conn_i32_bytes(n, H, W) bytes of int32 + 2 n + 64 bytes) as the kernels see them
inline imsegm::ConnWork conn_work_from(int32_t *base_i32, size_t cap_bytes, uint8_t *base_u8, size_t n)
{
    imsegm::ConnWork w;
    int32_t *b = base_i32;
    w.parent = b; b += n;
    w.csize = b; b += n;
    w.newlabel = b; b += n;
    w.adjptr = b; b += n;
    w.queue = b; b += n;
    w.list = b; b += n;
    w.slotmap = b; b += n;
    w.bbox = b; b += n;
    w.blocksum = b; b += (n / 4096) + 32;
    w.counters = b; b += 64;
    w.dense = b;
    w.dense_ints = (cap_bytes - (size_t)((unsigned char *)b - (unsigned char *)base_i32)) / 4;
    w.visited = base_u8;
    return w;
}

// the pieces of a session's connectivity scratch as the kernels see them
inline imsegm::ConnWork make_conn_work(imsegm_image2d *im)
{
    return conn_work_from(im->conn_i32.as<int32_t>(), im->conn_i32.cap, im->conn_u8.as<uint8_t>(), im->n);
}
