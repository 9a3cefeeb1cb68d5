// batch.hip -- several images of ONE size through the whole colour pipeline in one chain of launches.
//
// What it replaces: the reference segments the images of an experiment by mapping `segment_image_model` over a pool of worker
// processes (/root/reference/experiments_segmentation/run_segm_slic_model_graphcut.py:451-473, 505-514 through
// imsegm/utilities/experiments.py:392-403 WrapExecuteSequence) -- one call of
// segment_color2d_slic_features_model_graphcut (imsegm/pipelines.py:160-241) per image.  A 647 x 1024 image is 672 assignment
// workgroups: less than three per compute unit, every launch starts and drains within one generation of workgroups, and the
// ~56 dependent launches of an image are latency, not work (150 000 dispatches/s whatever is tried).
// Here B images share every launch: image b is blockIdx.z, and because every per-image buffer sits at the same offset of its
// image's slice of ONE arena, a kernel reaches image b by adding b * slice bytes to each of its pointers (common.h ZBatch).
// Per batch: B uploads, ~55 launches (not 55 * B), TWO host synchronisations (label counts after connectivity, the end), B
// downloads.  Results are those of imsegm_image2d_run_color image by image, bit for bit (tests/test_gpu_batch.py).
#include "session.h"

#include <string>

struct imsegm_batch2d {
    imsegm_ctx *ctx = nullptr;
    int B = 0, H = 0, W = 0;
    size_t n = 0;
    DevBuf arena;                    // B slices + the staging area of the batch
    size_t slice = 0;                // bytes per image (the stride of ZBatch)
    std::vector<unsigned char> key;  // the layout the arena was zeroed for
    int *fail_host = nullptr;        // page-locked word of the fused centroid update (one for the batch)
    void *pinned = nullptr;          // page-locked staging of the parameter blocks / counters
    size_t pinned_cap = 0;
    int last_n = 0;                  // images of the last run
    std::vector<int> n_labels;
};

namespace {

using namespace imsegm;

inline size_t al256(size_t v) { return (v + 255) & ~(size_t)255; }

// offsets of the per-image buffers inside a slice
struct Layout {
    size_t img, lab, nearest, labels, conn_i32, conn_i32_bytes, conn_u8, small, cent, tiles, feat, featK, seg, segm_out, total;
    // inside `seg` (the back half: parameters, graph, terms, cut)
    size_t o_misc, o_pw, o_sm, o_cl, o_sc, o_pc, o_mp, o_ld, o_lw, up_bytes;
    size_t d_proba, d_unary, d_unary_i, d_w, d_wi, d_edist, d_elen, d_edges, d_as, d_at, d_ar, d_ea, d_deg, d_dlow, d_es, d_wp, d_gl,
        d_lut, d_fstd, d_cacc, d_bitmap, d_cent, d_present, d_work, seg_bytes;
    int Kb, Ecap;                    // most labels an image can end up with, edge table rows
};

Layout make_layout(int H, int W, size_t elem, int K_grid, size_t n_tiles, long min_size, int C, int F)
{
    Layout L;
    memset(&L, 0, sizeof(L));        // (the bytes are compared: the key of the zeroed arena)
    const size_t n = (size_t)H * W;
    // connectivity keeps a component of >= min_size pixels: no more than n / min_size of them (+ the start label, + 1)
    L.Kb = (int)std::min<size_t>(n / (size_t)std::max<long>(min_size, 1) + 2, n + 2);
    L.Ecap = 3 * L.Kb + 64;          // planar graph of connected regions: E <= 3 K - 6
    size_t o = 0;
    auto take = [&](size_t bytes) { size_t at = o; o += al256(bytes); return at; };
    L.small = take(4096);
    L.img = take(n * 3 * elem + 64);
    L.lab = take(3 * n * sizeof(double));
    L.nearest = take(n * 4);
    L.labels = take(n * 4);
    L.conn_i32_bytes = conn_i32_bytes(n, H, W);
    L.conn_i32 = take(L.conn_i32_bytes);
    L.conn_u8 = take(2 * n + 64);
    L.cent = take(slic_cent_bytes(K_grid));
    L.tiles = take(slic_tiles_bytes(n_tiles, n));
    const size_t Kb = (size_t)L.Kb, E = (size_t)L.Ecap, words = (size_t)cdiv(L.Kb, 32);
    L.feat = take(Kb * (13 * 8 + 3 * 3 * 8 + 3 * 4) + 256);
    L.featK = take(Kb * (size_t)std::max(F, 1) * 8 + 64);
    L.segm_out = take(n * 4);
    // ---- the back half, relative to L.seg
    size_t q = 0;
    auto sub = [&](size_t bytes) { size_t at = q; q += (bytes + 63) & ~(size_t)63; return at; };
    const size_t FF = (size_t)F * F;
    L.o_misc = sub(256);             // K | E | status | gc status | energy (8) ... scalars[8] at +64
    L.o_pw = sub((size_t)C * C * 8);
    L.o_sm = sub((size_t)C * C * 4);
    L.o_cl = sub((size_t)C * 4);
    L.o_sc = sub((size_t)2 * F * 8);
    L.o_pc = sub((size_t)C * FF * 8);
    L.o_mp = sub((size_t)C * F * 8);
    L.o_ld = sub((size_t)C * 8);
    L.o_lw = sub((size_t)C * 8);
    L.up_bytes = q;
    q = al256(q);
    L.d_proba = sub(Kb * C * 8);
    L.d_unary = sub(Kb * C * 8);
    L.d_unary_i = sub(Kb * C * 4);
    L.d_w = sub(E * 8);
    L.d_wi = sub(E * 4);
    L.d_edist = sub(E * 8);
    L.d_elen = sub(E * 8);
    L.d_edges = sub(E * 8);
    L.d_as = sub((Kb + 1) * 4);
    L.d_at = sub(E * 8);
    L.d_ar = sub(E * 8);
    L.d_ea = sub(E * 8);
    L.d_deg = sub(Kb * 4);
    L.d_dlow = sub(Kb * 4);
    L.d_es = sub(Kb * 4);
    L.d_wp = sub(Kb * words * 4);
    L.d_gl = sub(Kb * 4);
    L.d_lut = sub(Kb * 4);
    L.d_fstd = sub((size_t)2 * std::max(F, 1) * 8);
    L.d_cacc = sub(Kb * 4 * 8);      // (right in front of the bitmap: one fill zeroes both)
    L.d_bitmap = sub(Kb * words * 4);
    L.d_cent = sub(Kb * 3 * 8);
    L.d_present = sub(Kb);
    L.d_work = sub(alpha_expansion_work_bytes(L.Kb, L.Ecap));
    L.seg_bytes = q;
    L.seg = take(L.seg_bytes + 256);
    L.total = al256(o);
    return L;
}

int batch_pinned(imsegm_batch2d *bt, size_t bytes)
{
    if (bytes <= bt->pinned_cap) return 0;
    if (bt->pinned) (void)hipHostFree(bt->pinned);
    bt->pinned = nullptr;
    bt->pinned_cap = 0;
    HIP_TRY(hipHostMalloc(&bt->pinned, bytes + 4096, hipHostMallocDefault));
    bt->pinned_cap = bytes + 4096;
    return 0;
}

}  // namespace

extern "C" {

int imsegm_batch2d_create(imsegm_ctx *ctx, int max_images, int height, int width, imsegm_batch2d **batch_out)
{
    if (bind(ctx)) return -1;
    if (!batch_out || max_images < 1 || max_images > 1024 || height <= 0 || width <= 0 || (long)height * width > 0x3fffffffL) {
        set_error("batch2d: 1..1024 images of a valid size");
        return -1;
    }
    imsegm_batch2d *bt = new imsegm_batch2d();
    bt->ctx = ctx;
    bt->B = max_images;
    bt->H = height;
    bt->W = width;
    bt->n = (size_t)height * width;
    bt->n_labels.assign(max_images, 0);
    *batch_out = bt;
    return 0;
}

void imsegm_batch2d_destroy(imsegm_batch2d *bt)
{
    if (!bt) return;
    (void)hipSetDevice(bt->ctx->device);
    (void)hipStreamSynchronize(bt->ctx->stream);
    bt->arena.release();
    if (bt->fail_host) (void)hipHostFree(bt->fail_host);
    if (bt->pinned) (void)hipHostFree(bt->pinned);
    delete bt;
}

int imsegm_batch2d_device_ptr(imsegm_batch2d *bt, int image, int which, void **ptr_out)
{
    if (!bt || bind(bt->ctx)) return -1;
    if (!ptr_out || image < 0 || image >= bt->last_n || bt->key.size() != sizeof(Layout)) {
        set_error("batch2d_device_ptr: no such image in the last batch");
        return -1;
    }
    Layout L;
    memcpy(&L, bt->key.data(), sizeof(L));
    unsigned char *base = bt->arena.as<unsigned char>() + (size_t)image * bt->slice;
    if (which == 0) *ptr_out = base + L.labels;
    else if (which == 1) *ptr_out = base + L.segm_out;
    else {
        set_error("batch2d_device_ptr: which = 0 (label map) or 1 (segmentation)");
        return -1;
    }
    return 0;
}

int imsegm_batch2d_run_color(imsegm_batch2d *bt, int n_images, const void *const *host_pixels, int dtype, int minmax_normalize,
                             int n_segments, double compactness, const double *taps, int radius, int max_iter, int start_label,
                             int feature_mask, const imsegm_gmm *gmm, int n_classes, const double *pairwise, int edge_type,
                             double edge_cost, int use_graphcut, const int32_t *classes_lut, int32_t *const *segm_out,
                             int *n_labels_out)
{
    if (!bt || bind(bt->ctx)) return -1;
    imsegm_ctx *ctx = bt->ctx;
    hipStream_t st = ctx->stream;
    const int H = bt->H, W = bt->W, C = n_classes;
    const size_t n = bt->n;
    if (n_images < 1 || n_images > bt->B || !host_pixels) {
        set_error("batch2d_run_color: between 1 and the batch's capacity of images");
        return -1;
    }
    const size_t es = dtype == IMSEGM_U8 ? 1 : dtype == IMSEGM_F32 ? 4 : dtype == IMSEGM_F64 ? 8 : 0;
    if (!es) {
        set_error("unsupported dtype");
        return -1;
    }
    if (!(compactness > 0) || n_segments < 1 || max_iter < 1 || (start_label != 0 && start_label != 1)) {
        set_error("slic: n_segments, compactness and max_iter must be positive, start_label 0 or 1");
        return -1;
    }
    if (feature_mask < 1 || feature_mask > 7 || !gmm || C < 1 || C > 16 || !pairwise) {
        set_error("batch2d_run_color: colour features (mask 1..7), a device class model, 1..16 classes and a pairwise matrix are required");
        return -1;
    }
    const int edge_code = edge_type & 0xff, spatial_norm = (edge_type & IMSEGM_EDGE_SPATIAL_NORM) ? 1 : 0;
    if (edge_code < 0 || edge_code > 5) {
        set_error("segment: unknown edge type");
        return -1;
    }
    const int nflags = ((feature_mask & 1) != 0) + ((feature_mask & 2) != 0) + ((feature_mask & 4) != 0);
    const int F = 3 * nflags;
    if (gmm->n_features != F || gmm->n_classes != C) {
        set_error("segment: class model does not match the resident features / number of classes");
        return -1;
    }
    for (int a = 0; a < C; ++a)
        for (int b = 0; b < C; ++b)
            if (pairwise[a * C + b] != pairwise[b * C + a]) {
                set_error("Cost matrix not square or not symmetric");
                return -1;
            }
    Taps tz, ty, tx;
    if (fill_taps(tz, taps, radius) || fill_taps(ty, taps, radius) || fill_taps(tx, taps, radius)) return -1;
    if (radius > 8 || knobs().pre_3pass) {
        set_error("batch2d_run_color: blur radius above 8 (sigma > 2) takes the single-image path");
        return -1;
    }

    // ---- geometry, layout, arena
    SlicState s;
    SlicGeometry geo;
    if (slic_geometry(H, W, n_segments, compactness, minmax_normalize, 0, 0, s, geo)) return -1;
    const int K = geo.K;
    const double segment_size = (double)n / (double)K;
    const long min_size = (long)(0.5 * segment_size), max_size = (long)(3.0 * segment_size);
    if (min_size < 1) {
        set_error("batch2d_run_color: superpixels of under two pixels take the single-image path");
        return -1;
    }
    const Layout L = make_layout(H, W, es, K, geo.n_tiles, min_size, C, F);
    const size_t slice = L.total;
    const size_t stage_row = al256(std::max<size_t>(L.up_bytes, 256));      // per image: parameter block in, counters out
    const size_t arena_bytes = slice * bt->B + stage_row * bt->B + 4096;
    const bool same = bt->key.size() == sizeof(L) && !memcmp(bt->key.data(), &L, sizeof(L)) && bt->slice == slice && bt->arena.cap >= arena_bytes;
    if (!same) {
        HIP_TRY(hipStreamSynchronize(st));
        if (bt->arena.ensure(arena_bytes)) return -1;
        // (the reduction words of every image start at zero and are left at zero by the kernels that use them)
        HIP_TRY(hipMemsetAsync(bt->arena.p, 0, bt->arena.cap, st));
        bt->slice = slice;
        bt->key.assign(reinterpret_cast<const unsigned char *>(&L), reinterpret_cast<const unsigned char *>(&L) + sizeof(L));
    }
    if (!bt->fail_host) HIP_TRY(hipHostMalloc((void **)&bt->fail_host, 64, hipHostMallocDefault));
    unsigned char *base = bt->arena.as<unsigned char>();
    unsigned char *stage_dev = base + slice * bt->B;
    ZBatch zb;
    zb.nz = n_images;
    zb.zs = slice;
    bt->last_n = n_images;

    // ---- uploads (a page-locked source travels by DMA when the stream gets there; a pageable one is staged by the runtime)
    bool all_pinned = true;
    for (int b = 0; b < n_images; ++b) {
        if (!host_pixels[b]) {
            set_error("batch2d_run_color: null image");
            return -1;
        }
        HIP_TRY(hipMemcpyAsync(base + (size_t)b * slice + L.img, host_pixels[b], n * 3 * es, hipMemcpyHostToDevice, st));
        all_pinned = all_pinned && is_pinned(host_pixels[b]);
    }
    if (!all_pinned) HIP_TRY(hipStreamSynchronize(st));

    // ---- SLIC: min / max, pre-processing, sweeps (image b = blockIdx.z everywhere)
    unsigned long long *keys = reinterpret_cast<unsigned long long *>(base + L.small);
    double *minmax = reinterpret_cast<double *>(keys + 2);
    double *premax = minmax + 2;
    double *lab = reinterpret_cast<double *>(base + L.lab);
    int32_t *nearest = reinterpret_cast<int32_t *>(base + L.nearest), *labels = reinterpret_cast<int32_t *>(base + L.labels);
    const int sp_all = ctx->begin(PG_SLIC);
    const int sp_pre = ctx->begin(PG_PRE);
    if (launch_minmax(base + L.img, dtype, n * 3, keys, minmax, st, premax, zb)) return -1;       // (and premax = 0)
    if (launch_preprocess_color2d(base + L.img, dtype, H, W, minmax_normalize, minmax, tz, ty, tx, 1.0 / compactness, lab, nullptr, premax,
                                  st, true, zb))
        return -1;
    ctx->end(sp_pre);
    slic_place_state(s, geo, base + L.cent, base + L.tiles, premax, bt->fail_host);
    ProfHook hook;
    if (ctx->profile) {
        hook.user = ctx;
        hook.begin = [](void *u, int g) { return static_cast<imsegm_ctx *>(u)->begin(g); };
        hook.end = [](void *u, int id) { static_cast<imsegm_ctx *>(u)->end(id); };
        hook.pair = [](void *u, int g, hipEvent_t *a, hipEvent_t *b) { static_cast<imsegm_ctx *>(u)->pair(g, a, b); };
    }
    bool fused_update = false;
    if (launch_slic_iterations(s, lab, nullptr, nearest, max_iter, 0, hook, st, &fused_update, zb)) return -1;

    // ---- connectivity: the tile path on all maps, ONE synchronisation for the label counts of the batch
    ConnWork w = conn_work_from(reinterpret_cast<int32_t *>(base + L.conn_i32), L.conn_i32_bytes, base + L.conn_u8, n);
    std::vector<int> &nl = bt->n_labels;
    for (int attempt = 0; attempt < 2; ++attempt) {
        if (attempt == 1) {
            // a contribution bypassed the arrival count of the fused centroid update somewhere in the batch (an uncovered pixel,
            // a tile without a list): the sweeps once more with separate finalize launches, which take every case
            slic_sweep_note_fallback();
            SlicState plain = s;
            plain.done = nullptr;
            if (launch_slic_iterations(plain, lab, nullptr, nearest, max_iter, 0, hook, st, nullptr, zb)) return -1;
        }
        const int spc = ctx->begin(PG_CONN);
        if (launch_enforce_connectivity_batch(nearest, H, W, min_size, max_size, start_label, w, labels, nl.data(), st, zb,
                                              reinterpret_cast<int32_t *>(stage_dev)))
            return -1;
        ctx->end(spc);
        if (!fused_update || *bt->fail_host == 0) break;
        fused_update = false;
    }
    for (int b = 0; b < n_images; ++b) {
        if (nl[b] >= 0) continue;
        // this map left the tile path (an oversize component, more local components than a tile has slots, ...): alone
        // through the general path, on its own slice
        unsigned char *sb = base + (size_t)b * slice;
        ConnWork wb = conn_work_from(reinterpret_cast<int32_t *>(sb + L.conn_i32), L.conn_i32_bytes, sb + L.conn_u8, n);
        int got = 0;
        if (launch_enforce_connectivity(reinterpret_cast<int32_t *>(sb + L.nearest), 1, H, W, min_size, max_size, start_label, wb,
                                        reinterpret_cast<int32_t *>(sb + L.labels), &got, st))
            return -1;
        nl[b] = got;
    }
    ctx->end(sp_all);
    int K_cap = 1;
    for (int b = 0; b < n_images; ++b) {
        if (nl[b] > L.Kb) {
            set_error("batch2d_run_color: more labels than the bound the buffers were sized for");
            return -1;
        }
        K_cap = std::max(K_cap, nl[b]);
        if (n_labels_out) n_labels_out[b] = nl[b];
    }
    const int words = cdiv(K_cap, 32);
    const int Ecap = std::min(L.Ecap, 3 * K_cap + 64);       // (rows in use of the edge tables, which are sized for L.Kb labels)

    // ---- descriptors: colour statistics of the uploaded pixels on the label maps, feature table K x F
    {
        unsigned char *fb = base + L.feat;
        long long *acc = reinterpret_cast<long long *>(fb); fb += (size_t)L.Kb * 13 * 8;
        double *d_mean = reinterpret_cast<double *>(fb); fb += (size_t)L.Kb * 3 * 8;
        double *d_energy = reinterpret_cast<double *>(fb); fb += (size_t)L.Kb * 3 * 8;
        double *d_var = reinterpret_cast<double *>(fb); fb += (size_t)L.Kb * 3 * 8;
        float *d_mean32 = reinterpret_cast<float *>(fb);
        double maxabs = 255.0;
        if (dtype != IMSEGM_U8) {
            // float images: the fixed-point scale of the sums follows the largest magnitude over the batch
            if (launch_minmax(base + L.img, dtype, n * 3, keys, minmax, st, nullptr, zb)) return -1;
            if (launch_copy_rows(stage_dev, 16, minmax, slice, 16, n_images, st)) return -1;
            std::vector<double> mm((size_t)2 * n_images);
            HIP_TRY(hipMemcpyAsync(mm.data(), stage_dev, (size_t)16 * n_images, hipMemcpyDeviceToHost, st));
            HIP_TRY(hipStreamSynchronize(st));
            maxabs = 0.0;
            for (double v : mm) maxabs = std::max(maxabs, fabs(v));
            if (!(maxabs < 1e300)) maxabs = 1e300;
        }
        const int sps = ctx->begin(PG_STATS);
        if (launch_color_stats(base + L.img, dtype, labels, H, W, K_cap, maxabs, (feature_mask & 2) != 0, acc, d_mean, d_energy, d_var,
                               d_mean32, st, 0, 0, 1.0, 1.0, -1, nullptr, zb))
            return -1;
        if (launch_features_assemble(d_mean, d_energy, d_var, K_cap, feature_mask, reinterpret_cast<double *>(base + L.featK), st, 0, 0, zb))
            return -1;
        ctx->end(sps);
    }

    // ---- parameter blocks: one per image (its label count in front), up in ONE transfer, scattered to the slices
    unsigned char *seg = base + L.seg;
    if (batch_pinned(bt, stage_row * n_images + 64)) return -1;
    unsigned char *host = static_cast<unsigned char *>(bt->pinned);
    int smax = 0;
    double pmax = -DBL_MAX;
    int metric = 0;
    {
        unsigned char *row = host;
        memset(row, 0, stage_row);
        memcpy(row + L.o_pw, pairwise, (size_t)C * C * 8);
        int32_t *si = reinterpret_cast<int32_t *>(row + L.o_sm);
        for (int i = 0; i < C * C; ++i) {
            si[i] = (int32_t)(pairwise[i] * 100);                 // pygco: smooth cost * 100, truncated
            smax = std::max(smax, std::abs(si[i]));
            pmax = std::max(pmax, pairwise[i]);
        }
        metric = smooth_is_metric(si, C);
        if (classes_lut) memcpy(row + L.o_cl, classes_lut, (size_t)C * 4);
        if (gmm->scaler_mean) memcpy(row + L.o_sc, gmm->scaler_mean, (size_t)F * 8);
        if (gmm->scaler_scale) memcpy(row + L.o_sc + (size_t)F * 8, gmm->scaler_scale, (size_t)F * 8);
        memcpy(row + L.o_pc, gmm->prec_chol, (size_t)C * F * F * 8);
        memcpy(row + L.o_mp, gmm->mu_proj, (size_t)C * F * 8);
        memcpy(row + L.o_ld, gmm->log_det, (size_t)C * 8);
        memcpy(row + L.o_lw, gmm->log_weights, (size_t)C * 8);
        for (int b = 0; b < n_images; ++b) {
            if (b) memcpy(host + (size_t)b * stage_row, host, stage_row);
            reinterpret_cast<int32_t *>(host + (size_t)b * stage_row + L.o_misc)[0] = nl[b];      // E, status words, energy: zero
        }
    }
    HIP_TRY(hipMemcpyAsync(stage_dev, host, stage_row * n_images, hipMemcpyHostToDevice, st));
    if (launch_copy_rows(seg, slice, stage_dev, stage_row, L.up_bytes, n_images, st)) return -1;
    int32_t *misc = reinterpret_cast<int32_t *>(seg + L.o_misc);
    int32_t *K_dev = misc, *E_dev = misc + 1, *status = misc + 2;
    long long *energy = reinterpret_cast<long long *>(seg + L.o_misc + 16);
    double *scalars = reinterpret_cast<double *>(seg + L.o_misc + 64);

    // ---- graph: bitmap + centres, then the symmetric CSR
    uint32_t *bitmap = reinterpret_cast<uint32_t *>(seg + L.d_bitmap);
    long long *cacc = reinterpret_cast<long long *>(seg + L.d_cacc);
    double *centres = reinterpret_cast<double *>(seg + L.d_cent);
    const int spg = ctx->begin(PG_GRAPH);
    if (launch_adjacency_bitmap(labels, H, W, K_cap, bitmap, cacc, centres, seg + L.d_present, st, zb)) return -1;
    int32_t *edges = reinterpret_cast<int32_t *>(seg + L.d_edges);
    if (launch_graph_csr(bitmap, K_dev, K_cap, words, reinterpret_cast<int32_t *>(seg + L.d_wp), reinterpret_cast<int32_t *>(seg + L.d_deg),
                         reinterpret_cast<int32_t *>(seg + L.d_dlow), reinterpret_cast<int32_t *>(seg + L.d_as),
                         reinterpret_cast<int32_t *>(seg + L.d_es), E_dev, Ecap, edges, reinterpret_cast<int32_t *>(seg + L.d_at),
                         reinterpret_cast<int32_t *>(seg + L.d_ar), reinterpret_cast<int32_t *>(seg + L.d_ea), st, zb))
        return -1;
    ctx->end(spg);

    // ---- class probabilities, unary / edge terms, integer energies
    TermsArgs a;
    memset(&a, 0, sizeof(a));
    a.zs = slice;
    a.Kp = K_dev; a.K_cap = K_cap; a.Ep = E_dev; a.edge_capacity = Ecap; a.F = F; a.C = C;
    a.features = reinterpret_cast<double *>(base + L.featK);
    a.gmm = 1;
    a.scaler_mean = gmm->scaler_mean ? reinterpret_cast<double *>(seg + L.o_sc) : nullptr;
    a.scaler_scale = gmm->scaler_scale ? reinterpret_cast<double *>(seg + L.o_sc) + F : nullptr;
    a.prec_chol = reinterpret_cast<double *>(seg + L.o_pc);
    a.mu_proj = reinterpret_cast<double *>(seg + L.o_mp);
    a.log_det = reinterpret_cast<double *>(seg + L.o_ld);
    a.log_w = reinterpret_cast<double *>(seg + L.o_lw);
    a.const_term = gmm->const_term;
    a.proba = reinterpret_cast<double *>(seg + L.d_proba);
    a.edge_type = edge_code; a.spatial_norm = spatial_norm; a.edge_cost = edge_cost;
    a.edges = edges; a.centres = centres; a.ndim = 2;
    a.edge_dist = reinterpret_cast<double *>(seg + L.d_edist); a.edge_len = reinterpret_cast<double *>(seg + L.d_elen);
    a.unary = reinterpret_cast<double *>(seg + L.d_unary); a.weights = reinterpret_cast<double *>(seg + L.d_w);
    a.pairwise = reinterpret_cast<double *>(seg + L.o_pw); a.pairwise_max = pmax;
    a.unary_i = reinterpret_cast<int32_t *>(seg + L.d_unary_i); a.weights_i = reinterpret_cast<int32_t *>(seg + L.d_wi);
    a.smooth_max = smax; a.status = status; a.scalars = scalars; a.fstd = reinterpret_cast<double *>(seg + L.d_fstd);
    const int spt = ctx->begin(PG_TERMS);
    if (launch_gc_terms(a, st, n_images)) return -1;
    ctx->end(spt);

    // ---- alpha-expansion: one workgroup per image of ONE launch (or the argmin of the unary cost for gc_regul <= 0)
    int32_t *glab = reinterpret_cast<int32_t *>(seg + L.d_gl);
    const int spc2 = ctx->begin(PG_GC);
    if (use_graphcut) {
        GcProblem p;
        p.K = K_cap; p.C = C; p.E = Ecap; p.E_dev = E_dev; p.K_dev = K_dev;
        p.edges = edges; p.w = a.weights_i; p.unary = a.unary_i; p.smooth = reinterpret_cast<int32_t *>(seg + L.o_sm);
        p.metric = metric;
        if (launch_alpha_expansion(p, reinterpret_cast<int32_t *>(seg + L.d_as), reinterpret_cast<int32_t *>(seg + L.d_at),
                                   reinterpret_cast<int32_t *>(seg + L.d_ar), reinterpret_cast<int32_t *>(seg + L.d_ea), -1, glab, energy,
                                   status + 1, seg + L.d_work, st, zb))
            return -1;
    } else if (launch_unary_argmin(a.unary, K_dev, K_cap, C, glab, st, zb)) {
        return -1;
    }
    ctx->end(spc2);

    // ---- gather classes_[graph_labels][slic], results down
    int32_t *lut = reinterpret_cast<int32_t *>(seg + L.d_lut);
    if (launch_label_lut(glab, K_dev, K_cap, classes_lut ? reinterpret_cast<int32_t *>(seg + L.o_cl) : nullptr, lut, st, zb)) return -1;
    const int spq = ctx->begin(PG_GATHER);
    if (launch_gather_labels(lut, labels, n, reinterpret_cast<int32_t *>(base + L.segm_out), st, zb)) return -1;
    ctx->end(spq);
    if (segm_out)
        for (int b = 0; b < n_images; ++b)
            if (segm_out[b])
                HIP_TRY(hipMemcpyAsync(segm_out[b], base + (size_t)b * slice + L.segm_out, n * 4, hipMemcpyDeviceToHost, st));
    if (launch_copy_rows(stage_dev, 16, misc, slice, 16, n_images, st)) return -1;
    std::vector<int32_t> hmisc((size_t)4 * n_images);
    HIP_TRY(hipMemcpyAsync(hmisc.data(), stage_dev, (size_t)16 * n_images, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    for (int b = 0; b < n_images; ++b) {
        const int32_t *m = hmisc.data() + (size_t)4 * b;
        if (m[2] & 2) {
            set_error("batch2d_run_color: image " + std::to_string(b) + " has more graph edges than a planar graph of its labels");
            return -1;
        }
        if (use_graphcut && (m[2] & 1)) {
            set_error("cut_general_graph: smoothness term is larger than GCO_MAX_ENERGYTERM");
            return -1;
        }
        if (use_graphcut && m[3] != 0) {
            set_error("alpha_expansion: max-flow did not converge");
            return -1;
        }
    }
    return 0;
}

}  // extern "C"
