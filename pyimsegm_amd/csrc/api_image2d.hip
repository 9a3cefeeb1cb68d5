// api_image2d.hip -- C ABI of the 2-D colour image session: SLIC, label maps, connectivity, label histograms, colour statistics,
// the graph call and the gathers (one of the files api.hip was split into in round 6; the helpers they share are declared in
// session.h)
#include "session.h"

// ---------------------------------------------------------------------------------------------------
// what of the 2-D SLIC state follows from the image size and the parameters (shared by api.hip and batch.hip):
// centroid grid (slic_superpixels.py _get_grid_centroids), integer steps (_slic.pyx), fp32 margin
// ---------------------------------------------------------------------------------------------------
int slic_geometry(int H, int W, int n_segments, double compactness, int minmax_normalize, int max_candidates, int slic_zero,
                  SlicState &s, SlicGeometry &geo)
{
    long shape[3] = { 1, H, W };
    GridAxis ax[3];
    regular_grid3(shape, n_segments, ax);
    long ny = 0, nx = 0;
    for (long y = ax[1].start; y < H; y += ax[1].step) ny++;
    for (long x = ax[2].start; x < W; x += ax[2].step) nx++;
    // (depth axis: one z = 0 plane, z start is always 0 for a length-1 axis)
    const int K = (int)(ny * nx);
    if (K < 1) {
        set_error("slic: empty centroid grid");
        return -1;
    }
    double fsteps[3];
    for (int i = 0; i < 3; ++i) fsteps[i] = ax[i].all ? 1.0 : (double)ax[i].step;
    float step = (float)std::max(fsteps[0], std::max(fsteps[1], fsteps[2]));
    GridAxis axk[3];
    regular_grid3(shape, K, axk);
    memset(&s, 0, sizeof(s));                              // (padding included: the bytes are the key of the cached graph)
    s.H = H; s.W = W; s.K = K;
    s.step_y = axk[1].all ? 1 : (int)axk[1].step;
    s.step_x = axk[2].all ? 1 : (int)axk[2].step;
    s.spatial_weight = 1.0 / ((double)step * (double)step);
    s.assign_units = 1;
    {
        // fp32 pre-selection margin (k_slic_assign): valid when the image entering rgb2lab lies in
        // [0, 1] (then |L|, |a|, |b| <= 108 before and after the convex blur), i.e. whenever the
        // min-max scaling is applied or the data already spans exactly [0, 1]
        const double u = 5.9604644775390625e-8;                      // 2^-24
        const double M = 108.0 * (1.0 / compactness) * 1.001;
        const double R = 2.0 * std::max(s.step_y, s.step_x) + 1.0;
        const double E = 3.0 * R + 64.0;
        const double G = sqrt(2.0 * s.spatial_weight) * E + sqrt(3.0) * 4.0 * M + 16.0;
        s.kappa = (float)(2.0 * u * (G + 1.0) * 1.0001);
        s.fast32 = (minmax_normalize != 0 && max_candidates >= 0 && s.kappa < 1e-2f && M < 4096.0) ? 1 : 0;
    }
    s.slico = slic_zero ? 1 : 0;
    s.grid_y0 = (int)ax[1].start; s.grid_dy = (int)ax[1].step;
    s.grid_x0 = (int)ax[2].start; s.grid_dx = (int)ax[2].step; s.grid_nx = (int)nx;
    geo.K = K;
    geo.n_tiles = (size_t)cdiv(W, SLIC_TILE_X) * cdiv(H, SLIC_TILE_Y);
    return 0;
}

// the pointers of the state into the centroid block (slic_cent_bytes) and the tile block (slic_tiles_bytes)
void slic_place_state(SlicState &s, const SlicGeometry &geo, unsigned char *cent, unsigned char *tiles, const double *premax, int *fail_host)
{
    const int K = geo.K;
    const size_t n_tiles = geo.n_tiles;
    s.premax = premax;
    unsigned char *cb = cent;
    s.acc = reinterpret_cast<long long *>(cb); cb += (size_t)K * 9 * 8;
    s.cy = reinterpret_cast<double *>(cb); cb += (size_t)K * 8;
    s.cx = reinterpret_cast<double *>(cb); cb += (size_t)K * 8;
    s.cL = reinterpret_cast<double *>(cb); cb += (size_t)K * 8;
    s.ca = reinterpret_cast<double *>(cb); cb += (size_t)K * 8;
    s.cb = reinterpret_cast<double *>(cb); cb += (size_t)K * 8;
    s.win = reinterpret_cast<int4 *>(cb); cb += (size_t)K * 16;
    s.mdc = reinterpret_cast<double *>(cb); cb += (size_t)K * 8;
    s.drift = reinterpret_cast<int *>(cb); cb += SLIC_DRIFT_SLOTS * sizeof(int);
    s.done = reinterpret_cast<int *>(cb); cb += (size_t)K * sizeof(int);
    s.fail_host = fail_host;
    s.tile_cands = reinterpret_cast<Cand *>(tiles);
    unsigned char *tb = tiles + n_tiles * SLIC_MAXC * sizeof(Cand);
    s.tile_rec = reinterpret_cast<Rec32 *>(tb); tb += n_tiles * SLIC_MAXC * sizeof(Rec32);
    s.tile_info = reinterpret_cast<TileInfo *>(tb); tb += n_tiles * sizeof(TileInfo);
    s.tile_k = reinterpret_cast<int *>(tb); tb += n_tiles * SLIC_MAXC * sizeof(int);
    s.tile_rows = reinterpret_cast<uint32_t *>(tb); tb += n_tiles * SLIC_MAXC * sizeof(uint32_t);
    s.tile_count = reinterpret_cast<int *>(tb);
    s.leftover_count = s.tile_count + n_tiles + 16;
    s.leftover = s.leftover_count + 16;
}


extern "C" {

int imsegm_image2d_slic(imsegm_image2d *im, int minmax_normalize, int n_segments, double compactness,
                        const double *taps_z, int radius_z, const double *taps_y, int radius_y,
                        const double *taps_x, int radius_x, int max_iter, int enforce_connectivity,
                        double min_size_factor, double max_size_factor, int start_label, int max_candidates,
                        int slic_zero, int *n_labels_out)
{
    if (!im || bind(im->ctx)) return -1;
    if (wrong_kind(im, false)) return -1;
    if (im->dtype < 0) {
        set_error("no image uploaded");
        return -1;
    }
    if (!(compactness > 0) || n_segments < 1 || max_iter < 1) {
        set_error("slic: n_segments, compactness and max_iter must be positive");
        return -1;
    }
    if (start_label != 0 && start_label != 1) {
        set_error("start_label should be 0 or 1.");
        return -1;
    }
    imsegm_ctx *ctx = im->ctx;
    hipStream_t st = ctx->stream;
    const int H = im->H, W = im->W;
    const size_t n = im->n;
    Taps tz, ty, tx;
    if (fill_taps(tz, taps_z, radius_z) || fill_taps(ty, taps_y, radius_y) || fill_taps(tx, taps_x, radius_x)) return -1;

    // centroid grid, steps, fp32 margin: everything of the SLIC state that follows from the sizes
    SlicState s;
    SlicGeometry geo;
    if (slic_geometry(H, W, n_segments, compactness, minmax_normalize, max_candidates, slic_zero, s, geo)) return -1;
    const int K = geo.K;
    const size_t n_tiles = geo.n_tiles;

    // buffers
    if (im->labA.ensure(3 * n * sizeof(double)) || im->labB.ensure(3 * n * sizeof(double))) return -1;
    if (im->nearest.ensure(n * 4) || im->labels.ensure(n * 4)) return -1;
    if (im->cent.ensure(slic_cent_bytes(K))) return -1;
    if (im->tiles.ensure(slic_tiles_bytes(n_tiles, n))) return -1;
    if (ensure_small(im)) return -1;

    unsigned long long *keys = im->small.as<unsigned long long>();
    double *minmax = reinterpret_cast<double *>(keys + 2);

    int sp_all = ctx->begin(PG_SLIC);
    int sp = ctx->begin(PG_PRE);
    double *premax = minmax + 2;                          // max |pre-processed value|, written on the device
    if (launch_minmax(im->img.p, im->dtype, n * 3, keys, minmax, st, premax)) return -1;      // (and premax = 0)
    if (launch_preprocess_color2d(im->img.p, im->dtype, H, W, minmax_normalize, minmax, tz, ty, tx, 1.0 / compactness,
                                  im->labA.as<double>(), im->labB.as<double>(), premax, st, true))
        return -1;
    ctx->end(sp);

    // profiling aids: read once per process (nothing of the hot path looks at the environment per image)
    static const int env_debug = getenv("IMSEGM_DEBUG_ASSIGN") ? atoi(getenv("IMSEGM_DEBUG_ASSIGN")) : 0;
    static const int env_units = getenv("IMSEGM_ASSIGN_UNITS") ? atoi(getenv("IMSEGM_ASSIGN_UNITS")) : 1;
    static const bool env_phase = getenv("IMSEGM_PHASE_PROF") != nullptr;
    s.debug = env_debug;
    s.assign_units = env_units;
    s.phase_prof = nullptr;
    static long long *phase_buf = nullptr;
    const size_t PHASE_SLOTS = 1 << 16;                    // workgroups of the assignment grid (profiling aid)
    if (env_phase) {
        if (!phase_buf) {
            HIP_TRY(hipMalloc(&phase_buf, PHASE_SLOTS * 32 * sizeof(long long)));
            HIP_TRY(hipMemset(phase_buf, 0, PHASE_SLOTS * 32 * sizeof(long long)));
        }
        s.phase_prof = phase_buf;
    }
    // arrival counters + the page-locked failure word of the centroid update inside the assignment kernel
    if (!im->slic_fail_host) HIP_TRY(hipHostMalloc((void **)&im->slic_fail_host, 64, hipHostMallocDefault));
    slic_place_state(s, geo, im->cent.as<unsigned char>(), im->tiles.as<unsigned char>(), premax, im->slic_fail_host);
    double *init_dev = nullptr;                            // the grid is generated on the device

    ProfHook hook;
    if (ctx->profile) {
        hook.user = ctx;
        hook.begin = [](void *u, int g) { return static_cast<imsegm_ctx *>(u)->begin(g); };
        hook.end = [](void *u, int id) { static_cast<imsegm_ctx *>(u)->end(id); };
        hook.pair = [](void *u, int g, hipEvent_t *a, hipEvent_t *b) { static_cast<imsegm_ctx *>(u)->pair(g, a, b); };
    }
    // Optional (IMSEGM_SLIC_GRAPH=1): the sweeps replayed from a captured HIP graph -- one submission instead of 31.  Measured
    // on ROCm 7.2 / MI355X it is SLOWER than the 31 plain launches (one image alone 1.97-2.03 ms against 1.86-1.87 ms; three
    // in flight 1.01-1.08 ms per image against 0.79-0.80 ms: the graph launches of different streams do not overlap the way
    // plain dispatches do), so it is off by default and kept for re-measuring on later runtimes.
    const bool use_graph = !ctx->profile && !s.phase_prof && knobs().slic_graph;
    bool used_persistent = false;          // the centroid update ran inside the assignment kernel: its failure word is read below
    if (use_graph) {
        struct SlicGraphKey {
            SlicState s;
            const double *lab;
            int32_t *labels;
            int max_iter, max_cand;
        } key;
        memset(&key, 0, sizeof(key));
        memcpy(&key.s, &s, sizeof(s));
        key.lab = im->labA.as<double>(); key.labels = im->nearest.as<int32_t>();
        key.max_iter = max_iter; key.max_cand = max_candidates;
        const bool same = im->slic_exec && im->slic_key.size() == sizeof(key) && !memcmp(im->slic_key.data(), &key, sizeof(key));
        if (!same) {
            if (im->slic_exec) {
                HIP_TRY(hipGraphExecDestroy(im->slic_exec));
                im->slic_exec = nullptr;
            }
            if (slic_prepare_device()) return -1;          // function attributes: not inside a capture
            hipGraph_t graph = nullptr;
            HIP_TRY(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
            SlicState captured = s;
            captured.done = nullptr;          // (a replayed graph has nobody to read the failure word: separate finalize launches)
            const int rc = launch_slic_iterations(captured, im->labA.as<double>(), init_dev, im->nearest.as<int32_t>(), max_iter,
                                                  max_candidates, hook, st);
            const hipError_t ec = hipStreamEndCapture(st, &graph);
            if (rc || ec != hipSuccess || !graph) {
                if (graph) (void)hipGraphDestroy(graph);
                if (!rc) set_error(std::string("slic: stream capture failed: ") + hipGetErrorString(ec));
                return -1;
            }
            const hipError_t ei = hipGraphInstantiate(&im->slic_exec, graph, nullptr, nullptr, 0);
            (void)hipGraphDestroy(graph);
            if (ei != hipSuccess) {
                im->slic_exec = nullptr;
                set_error(std::string("slic: graph instantiation failed: ") + hipGetErrorString(ei));
                return -1;
            }
            im->slic_key.assign(reinterpret_cast<unsigned char *>(&key), reinterpret_cast<unsigned char *>(&key) + sizeof(key));
        }
        HIP_TRY(hipGraphLaunch(im->slic_exec, st));
    } else {
        // (the failure word of the centroid update inside the assignment kernel is page-locked host memory, read after the next
        // synchronisation of this call: the connectivity stage ends with one)
        if (launch_slic_iterations(s, im->labA.as<double>(), init_dev, im->nearest.as<int32_t>(), max_iter, max_candidates, hook, st,
                                   &used_persistent))
            return -1;
    }

    if (s.phase_prof) {
        std::vector<long long> all((size_t)PHASE_SLOTS * 32);
        HIP_TRY(hipMemcpy(all.data(), s.phase_prof, all.size() * 8, hipMemcpyDeviceToHost));
        HIP_TRY(hipMemset(s.phase_prof, 0, all.size() * 8));
        bool any = false;
        for (size_t i = 15; i < all.size() && !any; i += 16) any = all[i] != 0;
        if (any && !knobs().phase_dump.empty()) {
            FILE *f = fopen(knobs().phase_dump.c_str(), "wb");
            if (f) {
                fwrite(all.data(), 8, all.size(), f);
                fclose(f);
            }
        }
        long long h[32] = { 0 };
        for (size_t i = 0; i < all.size(); ++i) h[i % 32] += all[i];
        for (int v = 0; v < 2; ++v) {
            const long long *q = h + v * 16;
            if (!q[15]) continue;
            fprintf(stderr, "[phase prof %s] waves=%lld  cycles/wave:", v ? "accum" : "last ", q[15]);
            for (int j = 0; j < 10; ++j) fprintf(stderr, " p%d=%.0f", j, (double)q[j] / (double)q[15]);
            fprintf(stderr, "\n");
        }
    }
    int n_labels = K + start_label;
    for (int attempt = 0; attempt < 2; ++attempt) {
        if (attempt == 1) {
            // the centroid update inside the assignment kernel gave the image back (a tile without a candidate list: more
            // candidates than a list holds): the sweeps again with separate finalize launches -- they take every case
            slic_sweep_note_fallback();
            SlicState plain = s;
            plain.done = nullptr;                              // separate finalize launches: they take every case
            if (launch_slic_iterations(plain, im->labA.as<double>(), init_dev, im->nearest.as<int32_t>(), max_iter, max_candidates, hook, st))
                return -1;
        }
        if (enforce_connectivity) {
            double segment_size = (double)n / (double)K;
            long min_size = (long)(min_size_factor * segment_size);
            long max_size = (long)(max_size_factor * segment_size);
            if (im->conn_i32.ensure(conn_i32_bytes(n, H, W)) || im->conn_u8.ensure(2 * n + 64)) return -1;
            ConnWork w = make_conn_work(im);
            int spc = ctx->begin(PG_CONN);
            // the raw assignment carries no start_label offset; the reference adds it before the
            // connectivity pass, which only matters through mask_label = start_label - 1 (no masked
            // pixels here), so the raw labels can be used as they are
            if (launch_enforce_connectivity(im->nearest.as<int32_t>(), 1, H, W, min_size, max_size, start_label, w,
                                            im->labels.as<int32_t>(), &n_labels, st))
                return -1;
            ctx->end(spc);
        } else {
            if (start_label != 0) {
                set_error("enforce_connectivity=False is only supported with start_label=0");
                return -1;
            }
            HIP_TRY(hipMemcpyAsync(im->labels.p, im->nearest.p, n * 4, hipMemcpyDeviceToDevice, st));
            if (used_persistent) HIP_TRY(hipStreamSynchronize(st));     // (the failure word is read below)
        }
        if (!used_persistent || *im->slic_fail_host == 0) break;       // (connectivity ended with a synchronisation)
        static const bool verbose = getenv("IMSEGM_DEBUG_SWEEPS") != nullptr;
        if (verbose) fprintf(stderr, "[slic sweeps] %d x %d, K = %d: handed back, code %d\n", H, W, K, *im->slic_fail_host);
        used_persistent = false;
    }
    ctx->end(sp_all);
    im->n_labels = n_labels;
    im->have_labels = true;
    im->labels_connected = false;
    im->graph_ready = false;
    if (n_labels_out) *n_labels_out = n_labels;
    return 0;
}

int imsegm_image2d_get_labels(imsegm_image2d *im, int64_t *labels_out)
{
    if (!im || bind(im->ctx)) return -1;
    if (!im->have_labels) {
        set_error("no label map");
        return -1;
    }
    std::vector<int32_t> tmp(im->n);
    HIP_TRY(hipMemcpyAsync(tmp.data(), im->labels.p, im->n * 4, hipMemcpyDeviceToHost, im->ctx->stream));
    HIP_TRY(hipStreamSynchronize(im->ctx->stream));
    for (size_t i = 0; i < im->n; ++i) labels_out[i] = tmp[i];
    return 0;
}

int imsegm_image2d_set_labels(imsegm_image2d *im, const int32_t *labels, int n_labels)
{
    if (!im || bind(im->ctx)) return -1;
    if (n_labels < 1) {
        set_error("n_labels must be positive");
        return -1;
    }
    if (im->labels.ensure(im->n * 4)) return -1;
    HIP_TRY(hipMemcpyAsync(im->labels.p, labels, im->n * 4, hipMemcpyHostToDevice, im->ctx->stream));
    HIP_TRY(hipStreamSynchronize(im->ctx->stream));
    im->n_labels = n_labels;
    im->have_labels = true;
    im->labels_connected = false;
    im->graph_ready = false;
    return 0;
}

// diagnostic: number of 2-D connectivity passes of this process that left the tile path for the general one
int imsegm_image2d_enforce_connectivity(imsegm_image2d *im, const int32_t *labels, long min_size, long max_size, int start_label,
                                        int *n_labels_out)
{
    if (!im || bind(im->ctx)) return -1;
    if (!labels || min_size < 0 || max_size < 1) {
        set_error("enforce_connectivity: label map, min_size >= 0 and max_size >= 1 are required");
        return -1;
    }
    if (start_label != 0 && start_label != 1) {
        set_error("start_label should be 0 or 1.");
        return -1;
    }
    hipStream_t st = im->ctx->stream;
    const size_t n = im->n;
    if (im->nearest.ensure(n * 4) || im->labels.ensure(n * 4)) return -1;
    if (im->conn_i32.ensure(conn_i32_bytes(n, im->D == 1 ? im->H : 0, im->D == 1 ? im->W : 0)) || im->conn_u8.ensure(2 * n + 64)) return -1;
    HIP_TRY(hipMemcpyAsync(im->nearest.p, labels, n * 4, hipMemcpyHostToDevice, st));
    ConnWork w = make_conn_work(im);
    int n_labels = 0;
    if (launch_enforce_connectivity(im->nearest.as<int32_t>(), im->D, im->H, im->W, min_size, max_size, start_label, w,
                                    im->labels.as<int32_t>(), &n_labels, st))
        return -1;
    im->n_labels = n_labels;
    im->have_labels = true;
    im->labels_connected = false;
    im->graph_ready = false;
    if (n_labels_out) *n_labels_out = n_labels;
    return 0;
}

// labeling.py:208-247 histogram_regions_labels_counts(slic, segm) on the resident label map (any session kind)
int imsegm_image2d_label_hist(imsegm_image2d *im, const int32_t *annot, int nb_annot, int64_t *hist_out)
{
    if (!im || bind(im->ctx)) return -1;
    if (!im->have_labels) {
        set_error("label_hist needs a label map (run slic or set_labels first)");
        return -1;
    }
    if (!annot || !hist_out || nb_annot < 1) {
        set_error("label_hist: annotation, output and a positive number of annotation labels are required");
        return -1;
    }
    const size_t bins = (size_t)im->n_labels * (size_t)nb_annot;
    if (bins > ((size_t)1 << 31)) {
        set_error("label_hist: histogram of more than 2^31 bins");
        return -1;
    }
    hipStream_t st = im->ctx->stream;
    if (im->annot.ensure(im->n * 4 + 32) || im->hist.ensure(bins * 8)) return -1;
    HIP_TRY(hipMemcpyAsync(im->annot.p, annot, im->n * 4, hipMemcpyHostToDevice, st));
    if (launch_label_hist(im->labels.as<int32_t>(), im->annot.as<int32_t>(), im->n, im->n_labels, nb_annot,
                          im->hist.as<unsigned long long>(), st))
        return -1;
    HIP_TRY(hipMemcpyAsync(hist_out, im->hist.p, bins * 8, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return 0;
}

int imsegm_image2d_get_lab(imsegm_image2d *im, double *lab_out)
{
    if (!im || bind(im->ctx)) return -1;
    if (wrong_kind(im, false)) return -1;
    if (im->labA.cap < 3 * im->n * 8) {
        set_error("slic has not been run");
        return -1;
    }
    HIP_TRY(hipMemcpyAsync(lab_out, im->labA.p, 3 * im->n * 8, hipMemcpyDeviceToHost, im->ctx->stream));
    HIP_TRY(hipStreamSynchronize(im->ctx->stream));
    return 0;
}

int imsegm_image2d_get_nearest(imsegm_image2d *im, int32_t *nearest_out)
{
    if (!im || bind(im->ctx)) return -1;
    if (im->nearest.cap < im->n * 4) {
        set_error("slic has not been run");
        return -1;
    }
    HIP_TRY(hipMemcpyAsync(nearest_out, im->nearest.p, im->n * 4, hipMemcpyDeviceToHost, im->ctx->stream));
    HIP_TRY(hipStreamSynchronize(im->ctx->stream));
    return 0;
}

int imsegm_image2d_color_stats(imsegm_image2d *im, double *mean_out, double *energy_out, double *var_out)
{
    if (!im || bind(im->ctx)) return -1;
    if (wrong_kind(im, false)) return -1;
    if (!im->have_labels || im->dtype < 0) {
        set_error("color_stats needs an uploaded image and a label map");
        return -1;
    }
    imsegm_ctx *ctx = im->ctx;
    hipStream_t st = ctx->stream;
    double maxabs = 255.0;
    if (im->dtype != IMSEGM_U8) {
        if (ensure_small(im)) return -1;
        unsigned long long *keys = im->small.as<unsigned long long>();
        double *minmax = reinterpret_cast<double *>(keys + 2);
        if (launch_minmax(im->img.p, im->dtype, im->n * 3, keys, minmax, st)) return -1;
        double mm[2];
        HIP_TRY(hipMemcpyAsync(mm, minmax, 16, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        maxabs = std::max(fabs(mm[0]), fabs(mm[1]));
        if (!(maxabs < 1e300)) maxabs = 1e300;
    }
    return stats_run(im, im->img.p, im->dtype, maxabs, 0, 0, 1.0, 1.0, mean_out, energy_out, var_out);
}

int imsegm_image2d_graph(imsegm_image2d *im, int32_t *edges_out, int edge_capacity, int *n_edges_out,
                         double *centres_out, uint8_t *present_out)
{
    if (!im || bind(im->ctx)) return -1;
    if (wrong_kind(im, false)) return -1;
    if (!im->have_labels) {
        set_error("graph needs a label map");
        return -1;
    }
    imsegm_ctx *ctx = im->ctx;
    hipStream_t st = ctx->stream;
    const int K = im->n_labels;
    if ((double)K * (double)K / 8.0 > 64e9) {         // K x K bitmap: 11 GB at the 3e5 supervoxels of config 5
        set_error("adjacency bitmap: too many labels (K*K/8 bytes must stay below 64 GB)");
        return -1;
    }
    if (edge_capacity < 0) edge_capacity = 0;
    size_t words = (size_t)cdiv(K, 32);
    size_t bytes = (size_t)K * words * 4 + (size_t)K * 3 * 8 + (size_t)edge_capacity * 8 + (size_t)K * 2 * 8 + (size_t)K * 4 + K + 512;
    if (im->graph.ensure(bytes)) return -1;
    unsigned char *b = im->graph.as<unsigned char>();
    long long *cacc = reinterpret_cast<long long *>(b); b += (size_t)K * 3 * 8;
    double *centres = reinterpret_cast<double *>(b); b += (size_t)K * 2 * 8;
    uint32_t *bitmap = reinterpret_cast<uint32_t *>(b); b += (size_t)K * words * 4;
    int32_t *edges = reinterpret_cast<int32_t *>(b); b += (size_t)edge_capacity * 8;
    int32_t *rowcount = reinterpret_cast<int32_t *>(b); b += (size_t)K * 4;
    int32_t *n_edges_dev = reinterpret_cast<int32_t *>(b); b += 16;
    uint8_t *present = b;
    int sp = ctx->begin(PG_GRAPH);
    if (launch_adjacency_centres(im->labels.as<int32_t>(), im->H, im->W, K, bitmap, cacc, edges, edge_capacity, n_edges_dev,
                                 centres, present, rowcount, st))
        return -1;
    ctx->end(sp);
    // one D2H of the whole result block through pinned memory: centres | edges | rowcount | n_edges | present
    size_t off_edges = (size_t)K * 16 + (size_t)K * words * 4;
    (void)off_edges;
    size_t sz_c = (size_t)K * 16, sz_e = (size_t)edge_capacity * 8, sz_p = (size_t)K;
    unsigned char *host = static_cast<unsigned char *>(ctx->stage(sz_c + sz_e + sz_p + 64));
    if (!host) {
        set_error("cannot allocate pinned staging memory");
        return -1;
    }
    HIP_TRY(hipMemcpyAsync(host, n_edges_dev, 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(host + 64, centres, sz_c, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(host + 64 + sz_c, present, sz_p, hipMemcpyDeviceToHost, st));
    if (edge_capacity > 0) HIP_TRY(hipMemcpyAsync(host + 64 + sz_c + sz_p, edges, sz_e, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    int ne = *reinterpret_cast<int *>(host);
    if (centres_out) memcpy(centres_out, host + 64, sz_c);
    if (present_out) memcpy(present_out, host + 64 + sz_c, sz_p);
    if (edges_out && ne > 0) memcpy(edges_out, host + 64 + sz_c + sz_p, (size_t)std::min(ne, edge_capacity) * 8);
    *n_edges_out = ne;
    return 0;
}

int imsegm_image2d_gather(imsegm_image2d *im, const int32_t *graph_labels, const double *proba, int n_classes,
                          int32_t *segm_out, double *soft_out)
{
    if (!im || bind(im->ctx)) return -1;
    if (!im->have_labels) {
        set_error("gather needs a label map");
        return -1;
    }
    imsegm_ctx *ctx = im->ctx;
    hipStream_t st = ctx->stream;
    const int K = im->n_labels;
    const size_t n = im->n;
    size_t lut_bytes = (size_t)K * 4 + 64 + (proba ? (size_t)K * n_classes * 8 : 0);
    if (im->gather_lut.ensure(lut_bytes)) return -1;
    double *d_proba = im->gather_lut.as<double>();
    int32_t *d_gl = reinterpret_cast<int32_t *>(im->gather_lut.as<unsigned char>() + (proba ? (size_t)K * n_classes * 8 : 0));
    if (proba && n_classes < 1) {
        set_error("n_classes must be positive");
        return -1;
    }
    if (graph_labels && im->gather_out_i.ensure(n * 4)) return -1;
    if (proba && im->gather_out_f.ensure(n * n_classes * 8)) return -1;
    // both LUTs travel in one pinned block: [proba K x C f64 | labels K i32]
    const size_t pb = proba ? (size_t)K * n_classes * 8 : 0, lb = graph_labels ? (size_t)K * 4 : 0;
    unsigned char *host = static_cast<unsigned char *>(ctx->stage(pb + lb + 64));
    if (!host) {
        set_error("cannot allocate pinned staging memory");
        return -1;
    }
    if (proba) memcpy(host, proba, pb);
    if (graph_labels) memcpy(host + pb, graph_labels, lb);
    HIP_TRY(hipMemcpyAsync(im->gather_lut.p, host, pb + lb, hipMemcpyHostToDevice, st));
    ctx->mark_stage_in_flight();
    int sp = ctx->begin(PG_GATHER);
    if (graph_labels && launch_gather_labels(d_gl, im->labels.as<int32_t>(), n, im->gather_out_i.as<int32_t>(), st)) return -1;
    if (proba && launch_gather_proba(d_proba, n_classes, im->labels.as<int32_t>(), n, im->gather_out_f.as<double>(), st)) return -1;
    ctx->end(sp);
    if (graph_labels && segm_out) HIP_TRY(hipMemcpyAsync(segm_out, im->gather_out_i.p, n * 4, hipMemcpyDeviceToHost, st));
    if (proba && soft_out) HIP_TRY(hipMemcpyAsync(soft_out, im->gather_out_f.p, n * n_classes * 8, hipMemcpyDeviceToHost, st));
    if ((graph_labels && segm_out) || (proba && soft_out)) HIP_TRY(hipStreamSynchronize(st));
    return 0;
}

int stats_run(imsegm_image2d *im, const void *src, int dtype, double maxabs, int planar, int prescale, double mul,
                     double div, double *mean_out, double *energy_out, double *var_out, long plane_stride)
{
    imsegm_ctx *ctx = im->ctx;
    hipStream_t st = ctx->stream;
    const int K = im->n_labels;
    size_t fb = (size_t)K * (13 * 8 + 3 * 3 * 8 + 3 * 4) + 256;
    if (im->feat.ensure(fb)) return -1;
    unsigned char *b = im->feat.as<unsigned char>();
    long long *acc = reinterpret_cast<long long *>(b); b += (size_t)K * 13 * 8;
    double *d_mean = reinterpret_cast<double *>(b); b += (size_t)K * 3 * 8;
    double *d_energy = reinterpret_cast<double *>(b); b += (size_t)K * 3 * 8;
    double *d_var = reinterpret_cast<double *>(b); b += (size_t)K * 3 * 8;
    float *d_mean32 = reinterpret_cast<float *>(b);
    int sp = ctx->begin(PG_STATS);
    if (launch_color_stats(src, dtype, im->labels.as<int32_t>(), im->H, im->W, K, maxabs, var_out != nullptr, acc, d_mean,
                           d_energy, d_var, d_mean32, st, planar, prescale, mul, div, plane_stride))
        return -1;
    ctx->end(sp);
    size_t ob = (size_t)K * 3 * 8;
    double *host = static_cast<double *>(ctx->stage(3 * ob));
    if (!host) {
        set_error("cannot allocate pinned staging memory");
        return -1;
    }
    HIP_TRY(hipMemcpyAsync(host, d_mean, 3 * ob, hipMemcpyDeviceToHost, st));     // mean | energy | var
    HIP_TRY(hipStreamSynchronize(st));
    if (mean_out) memcpy(mean_out, host, ob);
    if (energy_out) memcpy(energy_out, host + (size_t)K * 3, ob);
    if (var_out) memcpy(var_out, host + (size_t)K * 6, ob);
    return 0;
}


}  // extern "C"
