// api_volume.hip -- C ABI of the gray-volume session: upload, 3-D SLIC, measure.label, gray statistics, the graph call
// (one of the files api.hip was split into in round 6: the C ABI of include/imsegm_hip.h by stage; the helpers they share are
// declared in session.h)
#include "session.h"

extern "C" {

// ---------------------------------------------------------------------------------------------------
// gray volumes (D x H x W)
// ---------------------------------------------------------------------------------------------------
int imsegm_volume_create(imsegm_ctx *ctx, int depth, int height, int width, imsegm_image2d **vol_out)
{
    if (bind(ctx)) return -1;
    if (depth <= 0 || height <= 0 || width <= 0 || (long)depth * height * width > 0x40000000L) {
        set_error("bad volume size");
        return -1;
    }
    imsegm_image2d *im = new imsegm_image2d();
    im->ctx = ctx;
    im->is_volume = true;
    im->D = depth;
    im->H = height;
    im->W = width;
    im->n = (size_t)depth * height * width;
    *vol_out = im;
    return 0;
}

int imsegm_volume_upload(imsegm_image2d *im, const void *host_voxels, int dtype, double slic_offset, double slic_scale)
{
    if (!im || bind(im->ctx)) return -1;
    if (wrong_kind(im, true)) return -1;
    size_t es = dtype == IMSEGM_U8 ? 1 : dtype == IMSEGM_F32 ? 4 : dtype == IMSEGM_F64 ? 8 : 0;
    if (!es) {
        set_error("unsupported dtype");
        return -1;
    }
    if (im->img.ensure(im->n * es + 16)) return -1;
    HIP_TRY(hipMemcpyAsync(im->img.p, host_voxels, im->n * es, hipMemcpyHostToDevice, im->ctx->stream));
    HIP_TRY(hipStreamSynchronize(im->ctx->stream));
    im->dtype = dtype;
    im->vol_off = slic_offset;
    im->vol_scale = slic_scale;
    im->tex_ready = false;
    im->feat_mask = 0;            // (a recycled session: the feature table of the previous volume is not this one's)
    return 0;
}


int imsegm_volume_slic(imsegm_image2d *im, int n_segments, double compactness, const double *taps_z, int radius_z,
                       const double *taps_y, int radius_y, const double *taps_x, int radius_x, const double *spacing,
                       int max_iter, int enforce_connectivity, double min_size_factor, double max_size_factor,
                       int start_label, int *n_labels_out)
{
    if (!im || bind(im->ctx)) return -1;
    if (wrong_kind(im, true)) return -1;
    if (im->dtype < 0) {
        set_error("no volume uploaded");
        return -1;
    }
    if (!(compactness > 0) || n_segments < 1 || max_iter < 1 || !spacing) {
        set_error("slic: n_segments, compactness and max_iter must be positive");
        return -1;
    }
    if (start_label != 0 && start_label != 1) {
        set_error("start_label should be 0 or 1.");
        return -1;
    }
    imsegm_ctx *ctx = im->ctx;
    hipStream_t st = ctx->stream;
    const int D = im->D, H = im->H, W = im->W;
    const size_t n = im->n;
    Taps tz, ty, tx;
    if (fill_taps(tz, taps_z, radius_z) || fill_taps(ty, taps_y, radius_y) || fill_taps(tx, taps_x, radius_x)) return -1;
    long shape[3] = { D, H, W };
    GridAxis ax[3], axk[3];
    regular_grid3(shape, n_segments, ax);
    long cnt[3];
    for (int i = 0; i < 3; ++i) {
        cnt[i] = 0;
        for (long v = ax[i].start; v < shape[i]; v += ax[i].step) cnt[i]++;
    }
    const long Kl = cnt[0] * cnt[1] * cnt[2];
    if (Kl < 1 || Kl > 0x7fffffffL) {
        set_error("slic: bad centroid grid");
        return -1;
    }
    const int K = (int)Kl;
    double fs = 1.0;
    for (int i = 0; i < 3; ++i) fs = std::max(fs, ax[i].all ? 1.0 : (double)ax[i].step);
    float step = (float)fs;
    regular_grid3(shape, K, axk);
    if (im->labA.ensure(n * 8) || im->labB.ensure(n * 8) || im->nearest.ensure(n * 4) || im->labels.ensure(n * 4)) return -1;
    // a float32 volume stays float32 from end to end, as in scikit-image 0.18 (volume.hip, float32 section)
    const bool f32 = im->dtype == IMSEGM_F32;
    if (im->vol_cent.ensure((size_t)K * (4 * 8 + 6 * 4 + 6 * 8 + 4 * 4 + 6 * 4) + 256)) return -1;
    if (ensure_small(im)) return -1;
    double *premax = reinterpret_cast<double *>(im->small.as<unsigned char>() + 64);
    if (f32) {
        if (launch_vol_preprocess_f32(im->img.as<float>(), D, H, W, tz, ty, tx, 1.0 / compactness, im->labA.as<double>(),
                                      im->labB.as<double>(), st))
            return -1;
    } else {
        if (launch_vol_preprocess(im->img.p, im->dtype, im->vol_off, im->vol_scale, D, H, W, tz, ty, tx, 1.0 / compactness,
                                  im->labA.as<double>(), im->labB.as<double>(), st))
            return -1;
        if (launch_absmax_f64(im->labB.as<double>(), n, premax, st)) return -1;
    }
    VolState s;
    s.premax = premax;
    s.D = D; s.H = H; s.W = W; s.K = K;
    s.step_z = axk[0].all ? 1 : (int)axk[0].step;
    s.step_y = axk[1].all ? 1 : (int)axk[1].step;
    s.step_x = axk[2].all ? 1 : (int)axk[2].step;
    s.spatial_weight = 1.0 / ((double)step * (double)step);
    s.sz = spacing[0]; s.sy = spacing[1]; s.sx = spacing[2];
    unsigned char *cb = im->vol_cent.as<unsigned char>();
    s.cen = reinterpret_cast<double *>(cb); cb += (size_t)K * 4 * 8;
    s.acc = reinterpret_cast<long long *>(cb); cb += (size_t)K * 6 * 8;
    s.win = reinterpret_cast<int *>(cb); cb += (size_t)K * 6 * 4;
    s.cen32 = reinterpret_cast<float *>(cb); cb += (size_t)K * 4 * 4;
    s.bbox = reinterpret_cast<int *>(cb);
    for (int i = 0; i < 3; ++i) {
        s.grid_0[i] = (int)ax[i].start;
        s.grid_d[i] = (int)ax[i].step;
        s.grid_n[i] = (int)cnt[i];
    }
    {
        // brick lists: capacity = 4 x the expected number of windows meeting a brick, at least 64
        s.nbz = cdiv(D, VOL_BZ); s.nby = cdiv(H, VOL_BY); s.nbx = cdiv(W, VOL_BX);
        const size_t n_bricks = (size_t)s.nbz * s.nby * s.nbx;
        const double per_brick = (double)K / (double)n * (std::min(D, VOL_BZ) + 4.0 * s.step_z + 1) *
                                 (std::min(H, VOL_BY) + 4.0 * s.step_y + 1) * (std::min(W, VOL_BX) + 4.0 * s.step_x + 1);
        s.brick_cap = (int)std::min<double>(std::max(64.0, 4.0 * per_brick), (double)K);
        s.brick_cap = (s.brick_cap + 63) & ~63;
        if (knobs().brick_cap) s.brick_cap = std::max(1, knobs().brick_cap);   // (tests: overflow path)
        // (a float32 volume's lists hold whole entries -- 12 words: position, value, window, index -- so that the assignment kernel
        // reads what it needs of a candidate in one trip; 2.6 GB at the 65 536 bricks x 832 slots of BASELINE configs[4])
        const size_t words_per_slot = f32 ? 12 : 1;
        if (im->tiles.ensure((n_bricks + 64 + n_bricks * (size_t)s.brick_cap * words_per_slot) * sizeof(int) + 256)) return -1;
        s.brick_count = im->tiles.as<int>();
        s.brick_list = s.brick_count + ((n_bricks + 63) & ~(size_t)63);
        s.brick_entries = s.brick_list;                       // (16-byte aligned: n_bricks rounded to 64 words behind a hipMalloc)
    }
    int sp_all = ctx->begin(PG_SLIC);
    if (f32) {
        ProfHook hook;
        if (ctx->profile) {
            hook.user = ctx;
            hook.pair = [](void *u, int g, hipEvent_t *a, hipEvent_t *b) { static_cast<imsegm_ctx *>(u)->pair(g, a, b); };
        }
        if (launch_vol_slic_f32(s, im->labB.as<float>(), im->nearest.as<int32_t>(), max_iter, st, ctx->profile ? &hook : nullptr)) return -1;
    } else if (launch_vol_slic(s, im->labB.as<double>(), im->nearest.as<int32_t>(), max_iter, st)) {
        return -1;
    }
    int n_labels = K + start_label;
    if (enforce_connectivity) {
        double segment_size = (double)n / (double)K;
        long min_size = (long)(min_size_factor * segment_size);
        long max_size = (long)(max_size_factor * segment_size);
        // (a volume of one slice takes the 2-D tile path: its per-tile lists need room like those of an image)
        if (im->conn_i32.ensure(conn_i32_bytes(n, D == 1 ? H : 0, D == 1 ? W : 0)) || im->conn_u8.ensure(2 * n + 64)) return -1;
        ConnWork w = make_conn_work(im);
        if (launch_enforce_connectivity(im->nearest.as<int32_t>(), D, H, W, min_size, max_size, start_label, w,
                                        im->labels.as<int32_t>(), &n_labels, st))
            return -1;
    } else {
        if (start_label != 0) {
            set_error("enforce_connectivity=False is only supported with start_label=0");
            return -1;
        }
        HIP_TRY(hipMemcpyAsync(im->labels.p, im->nearest.p, n * 4, hipMemcpyDeviceToDevice, st));
    }
    ctx->end(sp_all);
    im->n_labels = n_labels;
    im->have_labels = true;
    im->labels_connected = enforce_connectivity != 0;
    im->graph_ready = false;
    if (n_labels_out) *n_labels_out = n_labels;
    return 0;
}

int imsegm_volume_label_cc(imsegm_image2d *im, int *n_labels_out)
{
    if (!im || bind(im->ctx)) return -1;
    if (wrong_kind(im, true)) return -1;
    if (!im->have_labels) {
        set_error("label_cc needs a label map");
        return -1;
    }
    hipStream_t st = im->ctx->stream;
    const size_t n = im->n;
    if (im->conn_i32.ensure(conn_i32_bytes(n)) || im->conn_u8.ensure(2 * n + 64)) return -1;
    ConnWork w = make_conn_work(im);
    if (im->labels_connected && !knobs().label_general && (size_t)im->n_labels <= n) {
        // the map the connectivity pass wrote: nothing to join, the labels are numbered by their first voxels (connectivity.hip)
        if (launch_label_connected(im->labels.as<int32_t>(), n, im->n_labels, w.newlabel, reinterpret_cast<uint32_t *>(w.visited), w.blocksum,
                                   w.counters + 16, w.counters, st))
            return -1;
    } else if (launch_label_cc(im->labels.as<int32_t>(), im->D, im->H, im->W, w.parent, w.newlabel, w.blocksum, w.counters, st)) {
        return -1;
    }
    int total = 0;
    HIP_TRY(hipMemcpyAsync(&total, w.counters, 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    im->n_labels = total + 1;          // 0 = background, components 1 .. total
    im->graph_ready = false;
    if (n_labels_out) *n_labels_out = im->n_labels;
    return 0;
}

int imsegm_volume_gray_stats(imsegm_image2d *im, double *mean_out, double *energy_out, double *var_out)
{
    if (!im || bind(im->ctx)) return -1;
    if (wrong_kind(im, true)) return -1;
    if (!im->have_labels || im->dtype < 0) {
        set_error("gray_stats needs an uploaded volume and a label map");
        return -1;
    }
    hipStream_t st = im->ctx->stream;
    double maxabs = 255.0;
    if (im->dtype != IMSEGM_U8) {
        if (ensure_small(im)) return -1;
        unsigned long long *keys = im->small.as<unsigned long long>();
        double *minmax = reinterpret_cast<double *>(keys + 2);
        if (launch_minmax(im->img.p, im->dtype, im->n, keys, minmax, st)) return -1;
        double mm[2];
        HIP_TRY(hipMemcpyAsync(mm, minmax, 16, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        maxabs = std::max(fabs(mm[0]), fabs(mm[1]));
        if (!(maxabs < 1e300)) maxabs = 1e300;
    }
    // the colour kernel with the single gray plane read as all three channels (plane stride 0), the
    // volume seen as a (D*H) x W image
    const int K = im->n_labels;
    std::vector<double> m((size_t)K * 3), e((size_t)K * 3), v((size_t)K * 3);
    const int H2 = im->D * im->H;
    int keepH = im->H;
    im->H = H2;
    int rc = stats_run(im, im->img.p, im->dtype, maxabs, 1, 0, 1.0, 1.0, mean_out ? m.data() : nullptr,
                       energy_out ? e.data() : nullptr, var_out ? v.data() : nullptr, 0);
    im->H = keepH;
    if (rc) return rc;
    for (int k = 0; k < K; ++k) {
        if (mean_out) mean_out[k] = m[(size_t)k * 3];
        if (energy_out) energy_out[k] = e[(size_t)k * 3];
        if (var_out) var_out[k] = v[(size_t)k * 3];
    }
    return 0;
}

int imsegm_volume_graph(imsegm_image2d *im, int32_t *edges_out, int edge_capacity, int *n_edges_out, double *centres_out,
                        uint8_t *present_out)
{
    if (!im || bind(im->ctx)) return -1;
    if (wrong_kind(im, true)) return -1;
    if (!im->have_labels) {
        set_error("graph needs a label map");
        return -1;
    }
    imsegm_ctx *ctx = im->ctx;
    hipStream_t st = ctx->stream;
    const int K = im->n_labels;
    if (edge_capacity < 0) edge_capacity = 0;
    size_t words = (size_t)cdiv(K, 32);
    // neighbours as bits of a K x K bitmap while that is small (one pass, no retry); beyond 256 MB -- K > 46 000; 11 GB at the 3 * 10^5
    // supervoxels of BASELINE configs[4] -- as a table of neighbour slots per label, widened until every row fits
    const bool as_table = (double)K * (double)words * 4.0 > 256e6 || knobs().adjacency_table;
    for (int cap = as_table ? 32 : 0;; cap *= 2) {
    const size_t store = as_table ? (size_t)K * cap * 4 : (size_t)K * words * 4;
    if (as_table && (cap > 65536 || store > 64e9)) {
        set_error("adjacency: a label with more than 65 536 neighbours of smaller number");
        return -1;
    }
    size_t bytes = store + (size_t)K * 4 * 8 + (size_t)edge_capacity * 8 + (size_t)K * 3 * 8 + (size_t)K * 4 + K + 512;
    if (im->graph.ensure(bytes)) return -1;
    unsigned char *b = im->graph.as<unsigned char>();
    long long *cacc = reinterpret_cast<long long *>(b); b += (size_t)K * 4 * 8;
    double *centres = reinterpret_cast<double *>(b); b += (size_t)K * 3 * 8;
    uint32_t *bitmap = reinterpret_cast<uint32_t *>(b); b += store;
    int32_t *edges = reinterpret_cast<int32_t *>(b); b += (size_t)edge_capacity * 8;
    int32_t *rowcount = reinterpret_cast<int32_t *>(b); b += (size_t)K * 4;
    int32_t *n_edges_dev = reinterpret_cast<int32_t *>(b); b += 16;             // [0] edges, [1] a row of the table was too narrow
    uint8_t *present = b;
    if (as_table) {
        int32_t *table = reinterpret_cast<int32_t *>(bitmap);
        if (launch_vol_adjacency_table(im->labels.as<int32_t>(), im->D, im->H, im->W, K, table, cap, n_edges_dev + 1, cacc, centres, present, st))
            return -1;
        int narrow = 0;
        HIP_TRY(hipMemcpyAsync(&narrow, n_edges_dev + 1, 4, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        if (narrow) continue;
        if (launch_edge_extract_table(table, K, cap, rowcount, edges, edge_capacity, n_edges_dev, st)) return -1;
    } else {
        if (launch_vol_adjacency(im->labels.as<int32_t>(), im->D, im->H, im->W, K, (int)words, bitmap, cacc, centres, present, st))
            return -1;
        if (launch_edge_extract(bitmap, K, (int)words, rowcount, edges, edge_capacity, n_edges_dev, st)) return -1;
    }
    int ne = 0;
    HIP_TRY(hipMemcpyAsync(&ne, n_edges_dev, 4, hipMemcpyDeviceToHost, st));
    if (centres_out) HIP_TRY(hipMemcpyAsync(centres_out, centres, (size_t)K * 24, hipMemcpyDeviceToHost, st));
    if (present_out) HIP_TRY(hipMemcpyAsync(present_out, present, K, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    if (edges_out && ne > 0) {
        HIP_TRY(hipMemcpyAsync(edges_out, edges, (size_t)std::min(ne, edge_capacity) * 8, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
    }
    *n_edges_out = ne;
    return 0;
    }
}


}  // extern "C"
