// api_fused.hip -- C ABI of the back half: the general graph cut, the resident feature table, the prepared graph and the fused segment call, the one-call colour pipeline
// (one of the files api.hip was split into in round 6: the C ABI of include/imsegm_hip.h by stage; the helpers they share are
// declared in session.h)
#include "session.h"

extern "C" {

int imsegm_cut_general_graph(imsegm_ctx *ctx, const int32_t *edges, int n_edges, const double *edge_weights,
                             const double *unary_cost, int n_sites, int n_labels, const double *pairwise_cost,
                             int n_iter, int32_t *labels_out, int64_t *energy_out)
{
    if (bind(ctx)) return -1;
    const int K = n_sites, C = n_labels, E = n_edges;
    if (K < 1 || C < 1 || E < 0) {
        set_error("cut_general_graph: bad sizes");
        return -1;
    }
    for (int j = 0; j < E; ++j) {
        int a = edges[2 * j], b = edges[2 * j + 1];
        if (a < 0 || b >= K || a >= b) {
            set_error("cut_general_graph: edges must satisfy 0 <= edges[:,0] < edges[:,1] < n_sites");
            return -1;
        }
    }
    for (int a = 0; a < C; ++a)
        for (int b = 0; b < C; ++b)
            if (pairwise_cost[a * C + b] != pairwise_cost[b * C + a]) {
                set_error("Cost matrix not square or not symmetric");
                return -1;
            }
    // device layout: [work | unary | w | smooth | edges | arc_start | arc_to | arc_rev | edge_arc | labels | energy | status];
    // the upload part (unary .. edge_arc) is assembled in ONE pinned host block with the same offsets
    hipStream_t st = ctx->stream;
    const size_t En = (size_t)std::max(E, 1);
    auto al = [](size_t v) { return (v + 63) & ~(size_t)63; };
    const size_t work_bytes = al(alpha_expansion_work_bytes(K, E));
    const size_t o_u = 0, o_w = o_u + al((size_t)K * C * 4), o_s = o_w + al(En * 4), o_e = o_s + al((size_t)C * C * 4);
    const size_t o_as = o_e + al(En * 8), o_at = o_as + al((size_t)(K + 1) * 4), o_ar = o_at + al(En * 8);
    const size_t o_ea = o_ar + al(En * 8), up_bytes = o_ea + al(En * 8);
    const size_t o_lab = up_bytes, o_en = o_lab + al((size_t)K * 4), o_st = o_en + 64, io_bytes = o_st + 64;
    if (ctx->gc_buf.ensure(work_bytes + io_bytes + 256)) return -1;
    unsigned char *host = static_cast<unsigned char *>(ctx->stage(io_bytes));
    if (!host) {
        set_error("cannot allocate pinned staging memory");
        return -1;
    }
    int32_t *ui = (int32_t *)(host + o_u), *wi = (int32_t *)(host + o_w), *si = (int32_t *)(host + o_s);
    int32_t *he = (int32_t *)(host + o_e), *arc_start = (int32_t *)(host + o_as), *arc_to = (int32_t *)(host + o_at);
    int32_t *arc_rev = (int32_t *)(host + o_ar), *edge_arc = (int32_t *)(host + o_ea);
    // pyGCO (gco/pygco.py): down_weight_factor and float -> int conversion (truncation)
    double mu = 0, mw = 0, mp = -DBL_MAX;
    for (size_t i = 0; i < (size_t)K * C; ++i) mu = std::max(mu, fabs(unary_cost[i]));
    for (int i = 0; i < E; ++i) mw = std::max(mw, fabs(edge_weights[i]));
    for (int i = 0; i < C * C; ++i) mp = std::max(mp, pairwise_cost[i]);
    double dwf = ((E > 0 && mw * mp > mu) ? mw * mp : mu) + 1e-10;
    for (size_t i = 0; i < (size_t)K * C; ++i) ui[i] = (int32_t)((unary_cost[i] / dwf) * 100000);
    for (int i = 0; i < E; ++i) wi[i] = (int32_t)((edge_weights[i] / dwf) * 1000);
    for (int i = 0; i < C * C; ++i) si[i] = (int32_t)(pairwise_cost[i] * 100);
    // GCO refuses energy terms above GCO_MAX_ENERGYTERM = 10000000
    int smax = 0;
    for (int i = 0; i < C * C; ++i) smax = std::max(smax, std::abs(si[i]));
    for (int i = 0; i < E; ++i)
        if ((long long)std::abs(wi[i]) * smax > 10000000LL) {
            set_error("cut_general_graph: smoothness term is larger than GCO_MAX_ENERGYTERM");
            return -1;
        }
    // CSR over directed arcs
    if (E > 0) memcpy(he, edges, (size_t)E * 8);
    for (int i = 0; i <= K; ++i) arc_start[i] = 0;
    for (int j = 0; j < E; ++j) {
        arc_start[edges[2 * j] + 1]++;
        arc_start[edges[2 * j + 1] + 1]++;
    }
    for (int i = 0; i < K; ++i) arc_start[i + 1] += arc_start[i];
    {
        std::vector<int32_t> fill(arc_start, arc_start + K);
        for (int j = 0; j < E; ++j) {
            int a = edges[2 * j], b = edges[2 * j + 1];
            int ia = fill[a]++, ib = fill[b]++;
            arc_to[ia] = b;
            arc_to[ib] = a;
            arc_rev[ia] = ib;
            arc_rev[ib] = ia;
            edge_arc[2 * j] = ia;
            edge_arc[2 * j + 1] = ib;
        }
    }
    unsigned char *dev = ctx->gc_buf.as<unsigned char>();
    void *work = dev;
    unsigned char *io = dev + work_bytes;
    int32_t *d_lab = (int32_t *)(io + o_lab);
    long long *d_energy = (long long *)(io + o_en);
    int32_t *d_status = (int32_t *)(io + o_st);
    HIP_TRY(hipMemcpyAsync(io, host, up_bytes, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemsetAsync(d_status, 0, 4, st));
    GcProblem p;
    p.K = K; p.C = C; p.E = E;
    p.edges = (int32_t *)(io + o_e); p.w = (int32_t *)(io + o_w); p.unary = (int32_t *)(io + o_u); p.smooth = (int32_t *)(io + o_s);
    p.metric = smooth_is_metric(si, C);
    int sp = ctx->begin(PG_GC);
    if (launch_alpha_expansion(p, (int32_t *)(io + o_as), (int32_t *)(io + o_at), (int32_t *)(io + o_ar), (int32_t *)(io + o_ea),
                               n_iter, d_lab, d_energy, d_status, work, st))
        return -1;
    ctx->end(sp);
    HIP_TRY(hipMemcpyAsync(host + o_lab, d_lab, io_bytes - o_lab, hipMemcpyDeviceToHost, st));   // labels | energy | status
    HIP_TRY(hipStreamSynchronize(st));
    memcpy(labels_out, host + o_lab, (size_t)K * 4);
    long long energy = *reinterpret_cast<long long *>(host + o_en);
    int32_t status = *reinterpret_cast<int32_t *>(host + o_st);
    if (status != 0) {
        set_error("alpha_expansion: max-flow did not converge");
        return -1;
    }
    if (energy_out) *energy_out = energy;
    return 0;
}


// ---------------------------------------------------------------------------------------------------
// fused back half of the pipeline: statistics -> feature table -> graph -> class model -> graph-cut terms ->
// alpha-expansion -> gathers, enqueued on the session's stream without a host round trip
// ---------------------------------------------------------------------------------------------------
// the placement of imsegm_image2d_features_place, consumed by the descriptor call that follows it: `own_F` columns at *col0 of a
// table *table_F wide (without a placement: the block is the table)
int take_placement(imsegm_image2d *im, int own_F, bool to_host, int *table_F, int *col0)
{
    *table_F = own_F;
    *col0 = 0;
    if (im->place_F <= 0) return 0;
    const int total = im->place_F, column = im->place_col;
    im->place_F = 0;
    if (column + own_F > total) {
        set_error("features_place: the columns of this descriptor group do not fit the table");
        return -1;
    }
    if (to_host && own_F != total) {
        set_error("features_place: a group placed into a wider table stays on the device (imsegm_image2d_get_features reads the table)");
        return -1;
    }
    *table_F = total;
    *col0 = column;
    return 0;
}

int imsegm_image2d_features_color(imsegm_image2d *im, int feature_mask, double *features_out)
{
    if (!im || bind(im->ctx)) return -1;
    if (!im->have_labels || im->dtype < 0) {
        set_error("features_color needs an uploaded image and a label map");
        return -1;
    }
    if (feature_mask < 1 || feature_mask > 7) {
        set_error("features_color: feature_mask is a combination of 1 (mean), 2 (std), 4 (energy)");
        return -1;
    }
    imsegm_ctx *ctx = im->ctx;
    hipStream_t st = ctx->stream;
    const int K = im->n_labels;
    double maxabs = 255.0;
    if (im->dtype != IMSEGM_U8) {
        if (ensure_small(im)) return -1;
        unsigned long long *keys = im->small.as<unsigned long long>();
        double *minmax = reinterpret_cast<double *>(keys + 2);
        if (launch_minmax(im->img.p, im->dtype, im->is_volume ? im->n : im->n * 3, keys, minmax, st)) return -1;
        double mm[2];
        HIP_TRY(hipMemcpyAsync(mm, minmax, 16, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        maxabs = std::max(fabs(mm[0]), fabs(mm[1]));
        if (!(maxabs < 1e300)) maxabs = 1e300;
    }
    // statistics without the D2H of stats_run
    size_t fb = (size_t)K * (13 * 8 + 3 * 3 * 8 + 3 * 4) + 256;
    if (im->feat.ensure(fb)) return -1;
    unsigned char *b = im->feat.as<unsigned char>();
    long long *acc = reinterpret_cast<long long *>(b); b += (size_t)K * 13 * 8;
    double *d_mean = reinterpret_cast<double *>(b); b += (size_t)K * 3 * 8;
    double *d_energy = reinterpret_cast<double *>(b); b += (size_t)K * 3 * 8;
    double *d_var = reinterpret_cast<double *>(b); b += (size_t)K * 3 * 8;
    float *d_mean32 = reinterpret_cast<float *>(b);
    int sp = ctx->begin(PG_STATS);
    int rc;
    if (im->is_volume)      // the gray plane read as all three channels (plane stride 0), the volume as a (D*H) x W image
        rc = launch_color_stats(im->img.p, im->dtype, im->labels.as<int32_t>(), im->D * im->H, im->W, K, maxabs, (feature_mask & 2) != 0,
                                acc, d_mean, d_energy, d_var, d_mean32, st, 1, 0, 1.0, 1.0, 0);
    else
        rc = launch_color_stats(im->img.p, im->dtype, im->labels.as<int32_t>(), im->H, im->W, K, maxabs, (feature_mask & 2) != 0, acc,
                                d_mean, d_energy, d_var, d_mean32, st, 0, 0, 1.0, 1.0, -1);
    if (rc) return -1;
    const int nflags = ((feature_mask & 1) != 0) + ((feature_mask & 2) != 0) + ((feature_mask & 4) != 0);
    const int F = 3 * nflags;
    int table_F = F, col0 = 0;
    if (take_placement(im, F, features_out != nullptr, &table_F, &col0)) return -1;
    if (im->featK.ensure((size_t)K * table_F * 8 + 64)) return -1;
    if (launch_features_assemble(d_mean, d_energy, d_var, K, feature_mask, im->featK.as<double>(), st, table_F, col0)) return -1;
    ctx->end(sp);
    im->feat_mask = table_F == F ? feature_mask : 8;          // (8: a table of several descriptor groups)
    im->feat_F = table_F;
    if (features_out) {
        HIP_TRY(hipMemcpyAsync(features_out, im->featK.p, (size_t)K * F * 8, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
    }
    return 0;
}

int imsegm_image2d_features_place(imsegm_image2d *im, int total_columns, int column)
{
    if (!im || bind(im->ctx)) return -1;
    if (total_columns < 1 || column < 0 || column >= total_columns) {
        set_error("features_place: 0 <= column < total_columns is required");
        return -1;
    }
    im->place_F = total_columns;
    im->place_col = column;
    return 0;
}

int imsegm_image2d_get_features(imsegm_image2d *im, double *features_out, int capacity_columns)
{
    if (!im || bind(im->ctx)) return -1;
    if (!im->have_labels || im->feat_mask == 0 || im->feat_F < 1) {
        set_error("get_features: no resident feature table");
        return -1;
    }
    if (!features_out || capacity_columns != im->feat_F) {
        set_error("get_features: the table has a different number of columns");
        return -1;
    }
    hipStream_t st = im->ctx->stream;
    HIP_TRY(hipMemcpyAsync(features_out, im->featK.p, (size_t)im->n_labels * im->feat_F * 8, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return 0;
}

// `edge_capacity` 0: sized for a planar adjacency graph (every superpixel connected); *edges_found: edges of the graph, also
// when the table was too small for them (return value -2: the caller retries with that many)
static int segment_impl(imsegm_image2d *im, const imsegm_gmm *gmm, const double *proba, int n_classes,
                        const double *pairwise, int edge_type, double edge_cost, int use_graphcut,
                        const int32_t *classes_lut, int32_t *segm_out, double *soft_out, int32_t *graph_labels_out,
                        double *proba_out, imsegm_terms_debug *debug_out, int edge_capacity, int *edges_found);

int imsegm_image2d_segment(imsegm_image2d *im, const imsegm_gmm *gmm, const double *proba, int n_classes,
                           const double *pairwise, int edge_type, double edge_cost, int use_graphcut,
                           const int32_t *classes_lut, int32_t *segm_out, double *soft_out, int32_t *graph_labels_out,
                           double *proba_out, imsegm_terms_debug *debug_out)
{
    int found = 0;
    int rc = segment_impl(im, gmm, proba, n_classes, pairwise, edge_type, edge_cost, use_graphcut, classes_lut, segm_out, soft_out,
                          graph_labels_out, proba_out, debug_out, 0, &found);
    // a label map whose regions are not connected (installed with imsegm_image2d_set_labels) can have more neighbour pairs than
    // a planar graph: once more with a table of the size the device has reported
    if (rc == -2)
        rc = segment_impl(im, gmm, proba, n_classes, pairwise, edge_type, edge_cost, use_graphcut, classes_lut, segm_out, soft_out,
                          graph_labels_out, proba_out, debug_out, found + 64, &found);
    return rc == -2 ? -1 : rc;
}

// ---- the graph of the resident label map: neighbour pairs + centre sums, then edges (a < b, ordered by (b, a)), CSR arcs in
// ascending neighbour order, reverse arcs, the edge -> arc table.  It depends on the label map only -- not on the class model --,
// so imsegm_image2d_graph_prepare may enqueue it ahead of the call that needs it (the volume pipeline: under the host's mixture fit).
static int default_edge_capacity(const imsegm_image2d *im, int K) { return im->is_volume ? 16 * K + 64 : 3 * K + 64; }   // planar: E <= 3K - 6

static GraphPlan graph_plan(const imsegm_image2d *im, int K, int edge_capacity)
{
    auto al = [](size_t v) { return (v + 63) & ~(size_t)63; };
    GraphPlan g;
    g.K = K;
    g.Ecap = edge_capacity > 0 ? edge_capacity : default_edge_capacity(im, K);
    g.words = cdiv(K, 32);
    // neighbours as bits of a K x K bitmap while that is small; a label volume beyond 256 MB of bitmap (K > 46 000; 11 GB at the
    // 3 * 10^5 supervoxels of BASELINE configs[4]) keeps them as 64 slots per label
    g.table = im->is_volume && ((double)K * (double)g.words * 4.0 > 256e6 || knobs().adjacency_table);
    g.cap = g.table ? 64 : 0;
    size_t d = 0;
    g.o_head = d; d += 64;                        // K | E | a row of the table was too narrow
    g.o_edges = d; d += al((size_t)g.Ecap * 8);
    g.o_as = d; d += al((size_t)(K + 1) * 4);
    g.o_at = d; d += al((size_t)g.Ecap * 8);
    g.o_ar = d; d += al((size_t)g.Ecap * 8);
    g.o_ea = d; d += al((size_t)g.Ecap * 8);
    g.o_deg = d; d += al((size_t)K * 4);
    g.o_dlow = d; d += al((size_t)K * 4);
    g.o_es = d; d += al((size_t)K * 4);
    g.o_cent = d; d += al((size_t)K * 3 * 8);
    g.o_present = d; d += al((size_t)K);
    g.o_store = d; d += al(g.table ? (size_t)K * g.cap * 4 : (size_t)K * g.words * 4);
    g.o_cacc = d; d += al((size_t)K * 4 * 8);     // (right behind the bitmap: one fill for both, graph.hip launch_adjacency_bitmap)
    g.o_wp = d; d += g.table ? 0 : al((size_t)K * g.words * 4);
    g.bytes = d + 256;
    return g;
}

__global__ void k_graph_head(int32_t *head, int K)
{
    head[0] = K;
    head[1] = 0;
    head[2] = 0;
}

// K_dev / E_dev: the words of the caller's parameter block, or null -> the head of the graph buffer itself (written by a kernel)
static int graph_enqueue(imsegm_image2d *im, const GraphPlan &g, int32_t *K_dev, int32_t *E_dev)
{
    imsegm_ctx *ctx = im->ctx;
    hipStream_t st = ctx->stream;
    if (im->gseg.ensure(g.bytes)) return -1;
    unsigned char *gb = im->gseg.as<unsigned char>();
    int32_t *head = reinterpret_cast<int32_t *>(gb + g.o_head);
    if (!K_dev || g.table) hipLaunchKernelGGL(k_graph_head, 1, 1, 0, st, head, g.K);
    if (!K_dev) {
        K_dev = head;
        E_dev = head + 1;
    }
    int32_t *store = reinterpret_cast<int32_t *>(gb + g.o_store);
    long long *cacc = reinterpret_cast<long long *>(gb + g.o_cacc);
    double *centres = reinterpret_cast<double *>(gb + g.o_cent);
    int sp = ctx->begin(PG_GRAPH);
    if (g.table) {
        if (launch_vol_adjacency_table(im->labels.as<int32_t>(), im->D, im->H, im->W, g.K, store, g.cap, head + 2, cacc, centres, gb + g.o_present, st))
            return -1;
        if (launch_graph_csr_table(store, K_dev, g.K, g.cap, head + 2, reinterpret_cast<int32_t *>(gb + g.o_deg), reinterpret_cast<int32_t *>(gb + g.o_dlow),
                                   reinterpret_cast<int32_t *>(gb + g.o_as), reinterpret_cast<int32_t *>(gb + g.o_es), E_dev, g.Ecap,
                                   reinterpret_cast<int32_t *>(gb + g.o_edges), reinterpret_cast<int32_t *>(gb + g.o_at),
                                   reinterpret_cast<int32_t *>(gb + g.o_ar), reinterpret_cast<int32_t *>(gb + g.o_ea), st))
            return -1;
    } else {
        uint32_t *bitmap = reinterpret_cast<uint32_t *>(store);
        if (im->is_volume) {
            if (launch_vol_adjacency(im->labels.as<int32_t>(), im->D, im->H, im->W, g.K, g.words, bitmap, cacc, centres, gb + g.o_present, st)) return -1;
        } else if (launch_adjacency_bitmap(im->labels.as<int32_t>(), im->H, im->W, g.K, bitmap, cacc, centres, gb + g.o_present, st)) {
            return -1;
        }
        if (launch_graph_csr(bitmap, K_dev, g.K, g.words, reinterpret_cast<int32_t *>(gb + g.o_wp), reinterpret_cast<int32_t *>(gb + g.o_deg),
                             reinterpret_cast<int32_t *>(gb + g.o_dlow), reinterpret_cast<int32_t *>(gb + g.o_as),
                             reinterpret_cast<int32_t *>(gb + g.o_es), E_dev, g.Ecap, reinterpret_cast<int32_t *>(gb + g.o_edges),
                             reinterpret_cast<int32_t *>(gb + g.o_at), reinterpret_cast<int32_t *>(gb + g.o_ar),
                             reinterpret_cast<int32_t *>(gb + g.o_ea), st))
            return -1;
    }
    ctx->end(sp);
    return 0;
}

// does the adjacency store of the fused path fit?  (the K x K bitmap and its word prefixes: what fits is asked of the device, not
// assumed -- the two arrays must fit the memory that is free NOW, plus what the session's own buffer already holds, with a tenth of
// the device left over; beyond that the caller builds the graph with imsegm_volume_graph and cuts it with
// imsegm_cut_general_graph.  Status IMSEGM_E_FUSED_PATH is what the host layer turns into that fall-back: ADVICE r4 / r5.)
static int graph_store_fits(imsegm_image2d *im, const GraphPlan &g)
{
    if (g.table) return 0;
    const double pair_bytes = 2.0 * (double)g.K * (double)g.words * 4.0;
    const int cap_mb = knobs().fused_bitmap_mb;
    if (cap_mb > 0 && pair_bytes > 1048576.0 * cap_mb) {
        set_error("segment: too many labels for the fused path (adjacency bitmap: IMSEGM_FUSED_BITMAP_MB)");
        return IMSEGM_E_FUSED_PATH;
    }
    if (pair_bytes > 16e6) {                  // (a 2-D image's graph: never in question, no query per image)
        size_t free_b = 0, total_b = 0;
        HIP_TRY(hipMemGetInfo(&free_b, &total_b));
        const double usable = (double)free_b + (double)im->gseg.cap - 0.1 * (double)total_b;
        if (pair_bytes > usable || pair_bytes > 48e9) {
            set_error("segment: too many labels for the fused path (adjacency bitmap: the device has no room for it)");
            return IMSEGM_E_FUSED_PATH;
        }
    }
    return 0;
}

int imsegm_image2d_graph_prepare(imsegm_image2d *im)
{
    if (!im || bind(im->ctx)) return -1;
    if (!im->have_labels) {
        set_error("graph_prepare needs a label map");
        return -1;
    }
    im->graph_ready = false;
    const GraphPlan g = graph_plan(im, im->n_labels, 0);
    if (int rc = graph_store_fits(im, g)) return rc;
    if (graph_enqueue(im, g, nullptr, nullptr)) return -1;
    im->gplan = g;
    im->graph_ready = true;
    return 0;
}

static int segment_impl(imsegm_image2d *im, const imsegm_gmm *gmm, const double *proba, int n_classes,
                        const double *pairwise, int edge_type, double edge_cost, int use_graphcut,
                        const int32_t *classes_lut, int32_t *segm_out, double *soft_out, int32_t *graph_labels_out,
                        double *proba_out, imsegm_terms_debug *debug_out, int edge_capacity, int *edges_found)
{
    if (!im || bind(im->ctx)) return -1;
    if (!im->have_labels) {
        set_error("segment needs a label map");
        return -1;
    }
    const int K = im->n_labels, C = n_classes;
    if (C < 1 || C > 16 || !pairwise || (!gmm && !proba)) {
        set_error("segment: 1..16 classes, a pairwise matrix and either a class model or probabilities are required");
        return -1;
    }
    int edge_code = edge_type & 0xff;
    const int spatial_norm = (edge_type & IMSEGM_EDGE_SPATIAL_NORM) ? 1 : 0;
    if (edge_code < 0 || edge_code > 5) {
        set_error("segment: unknown edge type");
        return -1;
    }
    const bool need_features = gmm != nullptr || edge_code == 5;
    if (need_features && im->feat_mask == 0) {
        set_error("segment: the class model / edge type needs the resident feature table (imsegm_image2d_features_color)");
        return -1;
    }
    const int F = need_features ? im->feat_F : 0;
    if (gmm && (gmm->n_features != F || gmm->n_classes != C)) {
        set_error("segment: class model does not match the resident features / number of classes");
        return -1;
    }
    for (int a = 0; a < C; ++a)
        for (int b = 0; b < C; ++b)
            if (pairwise[a * C + b] != pairwise[b * C + a]) {
                set_error("Cost matrix not square or not symmetric");
                return -1;
            }
    // the graph: prepared ahead (imsegm_image2d_graph_prepare, same label map, room for the edges asked for) or built here
    const bool prepared = im->graph_ready && im->gplan.K == K && (edge_capacity <= 0 || im->gplan.Ecap >= edge_capacity);
    const GraphPlan g = prepared ? im->gplan : graph_plan(im, K, edge_capacity);
    im->graph_ready = false;                   // (one segmentation per prepared graph: the cut works on the arcs' buffers)
    if (!prepared)
        if (int rc = graph_store_fits(im, g)) return rc;
    imsegm_ctx *ctx = im->ctx;
    hipStream_t st = ctx->stream;
    const size_t n = im->n;
    const int ndim = im->is_volume ? 3 : 2;
    const int Ecap = g.Ecap;
    auto al = [](size_t v) { return (v + 63) & ~(size_t)63; };
    // ---- host -> device parameter block (one pinned staging copy)
    const size_t FF = (size_t)F * F;
    size_t o = 0;
    const size_t o_misc = o; o += 256;          // K | E | status | pad | energy (8) | scalars[8]: initialised by the same copy
    const size_t o_pw = o; o += al((size_t)C * C * 8);
    const size_t o_sm = o; o += al((size_t)C * C * 4);
    const size_t o_cl = o; o += al((size_t)C * 4);
    const size_t o_sc = o; o += al((size_t)2 * F * 8);
    const size_t o_pc = o; o += al((size_t)C * FF * 8);
    const size_t o_mp = o; o += al((size_t)C * F * 8);
    const size_t o_ld = o; o += al((size_t)C * 8);
    const size_t o_lw = o; o += al((size_t)C * 8);
    const size_t o_pr = o; o += (gmm ? 0 : al((size_t)K * C * 8));
    const size_t up_bytes = o;
    // ---- device layout
    const size_t d_par = 0;
    size_t d = al(up_bytes);
    if (gmm) { /* proba lives behind the parameters */ }
    const size_t d_proba = gmm ? d : d_par + o_pr; if (gmm) d += al((size_t)K * C * 8);
    const size_t d_unary = d; d += al((size_t)K * C * 8);
    const size_t d_unary_i = d; d += al((size_t)K * C * 4);
    const size_t d_w = d; d += al((size_t)Ecap * 8);
    const size_t d_wi = d; d += al((size_t)Ecap * 4);
    const size_t d_edist = d; d += al((size_t)Ecap * 8);
    const size_t d_elen = d; d += al((size_t)Ecap * 8);
    const size_t d_gl = d; d += al((size_t)K * 4);
    const size_t d_lut = d; d += al((size_t)K * 4);
    const size_t d_misc = d_par + o_misc;
    const size_t d_fstd = d; d += al((size_t)2 * std::max(F, 1) * 8);
    const size_t d_work = d; d += al(alpha_expansion_work_bytes(K, Ecap));
    if (im->seg.ensure(d + 256)) return -1;
    unsigned char *dev = im->seg.as<unsigned char>();
    unsigned char *host = static_cast<unsigned char *>(ctx->stage(up_bytes + 64));
    if (!host) {
        set_error("cannot allocate pinned staging memory");
        return -1;
    }
    memset(host, 0, up_bytes);
    reinterpret_cast<int32_t *>(host + o_misc)[0] = K;            // E = 0 | status = 0 | gc status = 0 | energy = 0 behind it
    memcpy(host + o_pw, pairwise, (size_t)C * C * 8);
    int32_t *si = reinterpret_cast<int32_t *>(host + o_sm);
    int smax = 0;
    double pmax = -DBL_MAX;
    for (int i = 0; i < C * C; ++i) {
        si[i] = (int32_t)(pairwise[i] * 100);                 // pygco: smooth cost * 100, truncated
        smax = std::max(smax, std::abs(si[i]));
        pmax = std::max(pmax, pairwise[i]);
    }
    const int metric = smooth_is_metric(si, C);
    if (classes_lut) memcpy(host + o_cl, classes_lut, (size_t)C * 4);
    if (gmm) {
        if (gmm->scaler_mean) memcpy(host + o_sc, gmm->scaler_mean, (size_t)F * 8);
        if (gmm->scaler_scale) memcpy(host + o_sc + (size_t)F * 8, gmm->scaler_scale, (size_t)F * 8);
        memcpy(host + o_pc, gmm->prec_chol, (size_t)C * FF * 8);
        memcpy(host + o_mp, gmm->mu_proj, (size_t)C * F * 8);
        memcpy(host + o_ld, gmm->log_det, (size_t)C * 8);
        memcpy(host + o_lw, gmm->log_weights, (size_t)C * 8);
    } else {
        memcpy(host + o_pr, proba, (size_t)K * C * 8);
    }
    HIP_TRY(hipMemcpyAsync(dev + d_par, host, up_bytes, hipMemcpyHostToDevice, st));
    ctx->mark_stage_in_flight();
    int32_t *misc = reinterpret_cast<int32_t *>(dev + d_misc);
    int32_t *K_dev = misc, *E_dev = misc + 1, *status = misc + 2;
    long long *energy = reinterpret_cast<long long *>(dev + d_misc + 16);
    double *scalars = reinterpret_cast<double *>(dev + d_misc + 64);
    // ---- graph: neighbour pairs + centres, then the symmetric CSR (im->gseg)
    if (prepared) E_dev = nullptr;                           // (the prepared graph counted its edges in its own head)
    else if (graph_enqueue(im, g, K_dev, E_dev)) return -1;
    unsigned char *gb = im->gseg.as<unsigned char>();
    int32_t *ghead = reinterpret_cast<int32_t *>(gb + g.o_head);
    if (!E_dev) E_dev = ghead + 1;
    int32_t *edges = reinterpret_cast<int32_t *>(gb + g.o_edges);
    double *centres = reinterpret_cast<double *>(gb + g.o_cent);
    int32_t *arc_start = reinterpret_cast<int32_t *>(gb + g.o_as), *arc_to = reinterpret_cast<int32_t *>(gb + g.o_at);
    int32_t *arc_rev = reinterpret_cast<int32_t *>(gb + g.o_ar), *edge_arc = reinterpret_cast<int32_t *>(gb + g.o_ea);
    // ---- class probabilities, unary / edge terms, integer energies
    TermsArgs a;
    memset(&a, 0, sizeof(a));
    a.Kp = K_dev; a.K_cap = K; a.Ep = E_dev; a.edge_capacity = Ecap; a.F = F; a.C = C;
    a.features = need_features ? im->featK.as<double>() : nullptr;
    a.gmm = gmm ? 1 : 0;
    if (gmm) {
        a.scaler_mean = gmm->scaler_mean ? reinterpret_cast<double *>(dev + d_par + o_sc) : nullptr;
        a.scaler_scale = gmm->scaler_scale ? reinterpret_cast<double *>(dev + d_par + o_sc) + F : nullptr;
        a.prec_chol = reinterpret_cast<double *>(dev + d_par + o_pc);
        a.mu_proj = reinterpret_cast<double *>(dev + d_par + o_mp);
        a.log_det = reinterpret_cast<double *>(dev + d_par + o_ld);
        a.log_w = reinterpret_cast<double *>(dev + d_par + o_lw);
        a.const_term = gmm->const_term;
    }
    a.proba = reinterpret_cast<double *>(dev + d_proba);
    a.edge_type = edge_code; a.spatial_norm = spatial_norm; a.edge_cost = edge_cost;
    a.edges = edges; a.centres = centres; a.ndim = ndim;
    a.edge_dist = reinterpret_cast<double *>(dev + d_edist); a.edge_len = reinterpret_cast<double *>(dev + d_elen);
    a.unary = reinterpret_cast<double *>(dev + d_unary); a.weights = reinterpret_cast<double *>(dev + d_w);
    a.pairwise = reinterpret_cast<double *>(dev + d_par + o_pw); a.pairwise_max = pmax;
    a.unary_i = reinterpret_cast<int32_t *>(dev + d_unary_i); a.weights_i = reinterpret_cast<int32_t *>(dev + d_wi);
    a.smooth_max = smax; a.status = status; a.scalars = scalars; a.fstd = reinterpret_cast<double *>(dev + d_fstd);
    int spt = ctx->begin(PG_TERMS);
    if (launch_gc_terms(a, st)) return -1;
    ctx->end(spt);
    // ---- alpha-expansion (or the argmin of the unary cost for gc_regul <= 0)
    int32_t *glab = reinterpret_cast<int32_t *>(dev + d_gl);
    int spg = ctx->begin(PG_GC);
    if (use_graphcut) {
        GcProblem p;
        p.K = K; p.C = C; p.E = Ecap; p.E_dev = E_dev;
        p.edges = edges; p.w = a.weights_i; p.unary = a.unary_i; p.smooth = reinterpret_cast<int32_t *>(dev + d_par + o_sm);
        p.metric = metric;
        if (launch_alpha_expansion(p, arc_start, arc_to, arc_rev, edge_arc, -1, glab, energy, status + 1, dev + d_work, st))
            return -1;
    } else if (launch_unary_argmin(a.unary, K_dev, K, C, glab, st)) {
        return -1;
    }
    ctx->end(spg);
    // ---- gathers: classes_[graph_labels][slic] and proba[slic]
    int32_t *lut = reinterpret_cast<int32_t *>(dev + d_lut);
    if (launch_label_lut(glab, K_dev, K, classes_lut ? reinterpret_cast<int32_t *>(dev + d_par + o_cl) : nullptr, lut, st)) return -1;
    if (im->gather_out_i.ensure(n * 4)) return -1;
    const bool want_soft = soft_out != nullptr || (debug_out && debug_out->keep_soft_on_device);
    if (want_soft && im->gather_out_f.ensure(n * C * 8)) return -1;
    int spq = ctx->begin(PG_GATHER);
    if (launch_gather_labels(lut, im->labels.as<int32_t>(), n, im->gather_out_i.as<int32_t>(), st)) return -1;
    if (want_soft && launch_gather_proba(a.proba, C, im->labels.as<int32_t>(), n, im->gather_out_f.as<double>(), st)) return -1;
    ctx->end(spq);
    // ---- results (int32 / float64 as the reference returns them, or the narrow formats the caller asked for)
    const bool segm_u8 = debug_out && debug_out->segm_u8, soft_f32 = debug_out && debug_out->soft_f32;
    if ((segm_u8 && segm_out) || (soft_f32 && soft_out)) {
        const size_t off_soft = (n + 255) & ~(size_t)255;
        if (im->narrow.ensure(off_soft + n * C * 4 + 64)) return -1;
        unsigned char *nb = im->narrow.as<unsigned char>();
        if (segm_u8 && segm_out) {
            if (launch_narrow_labels_u8(im->gather_out_i.as<int32_t>(), nb, n, st)) return -1;
            HIP_TRY(hipMemcpyAsync(segm_out, nb, n, hipMemcpyDeviceToHost, st));
            segm_out = nullptr;
        }
        if (soft_f32 && soft_out) {
            float *f32 = reinterpret_cast<float *>(nb + off_soft);
            if (launch_narrow_soft_f32(im->gather_out_f.as<double>(), f32, n * C, st)) return -1;
            HIP_TRY(hipMemcpyAsync(soft_out, f32, n * C * 4, hipMemcpyDeviceToHost, st));
            soft_out = nullptr;
        }
    }
    if (segm_out) HIP_TRY(hipMemcpyAsync(segm_out, im->gather_out_i.p, n * 4, hipMemcpyDeviceToHost, st));
    if (soft_out) HIP_TRY(hipMemcpyAsync(soft_out, im->gather_out_f.p, n * C * 8, hipMemcpyDeviceToHost, st));
    if (graph_labels_out) HIP_TRY(hipMemcpyAsync(graph_labels_out, glab, (size_t)K * 4, hipMemcpyDeviceToHost, st));
    if (proba_out) HIP_TRY(hipMemcpyAsync(proba_out, a.proba, (size_t)K * C * 8, hipMemcpyDeviceToHost, st));
    int32_t hmisc[4] = { 0, 0, 0, 0 }, hgraph[4] = { 0, 0, 0, 0 };
    HIP_TRY(hipMemcpyAsync(hmisc, misc, sizeof(hmisc), hipMemcpyDeviceToHost, st));
    if (prepared || g.table) HIP_TRY(hipMemcpyAsync(hgraph, ghead, sizeof(hgraph), hipMemcpyDeviceToHost, st));
    if (debug_out) {
        if (debug_out->unary) HIP_TRY(hipMemcpyAsync(debug_out->unary, a.unary, (size_t)K * C * 8, hipMemcpyDeviceToHost, st));
        if (debug_out->unary_int) HIP_TRY(hipMemcpyAsync(debug_out->unary_int, a.unary_i, (size_t)K * C * 4, hipMemcpyDeviceToHost, st));
        if (debug_out->centres) HIP_TRY(hipMemcpyAsync(debug_out->centres, centres, (size_t)K * ndim * 8, hipMemcpyDeviceToHost, st));
        if (debug_out->energy) HIP_TRY(hipMemcpyAsync(debug_out->energy, energy, 8, hipMemcpyDeviceToHost, st));
    }
    HIP_TRY(hipStreamSynchronize(st));
    const int E = prepared ? hgraph[1] : hmisc[1];
    if (g.table && hgraph[2]) {
        set_error("segment: too many labels for the fused path (a label with more than 64 neighbours in the neighbour table)");
        return IMSEGM_E_FUSED_PATH;
    }
    if (debug_out) {
        debug_out->n_edges = E;
        const int Ec = std::min(E, debug_out->edge_capacity);
        if (Ec > 0) {
            if (debug_out->edges) HIP_TRY(hipMemcpy(debug_out->edges, edges, (size_t)Ec * 8, hipMemcpyDeviceToHost));
            if (debug_out->edge_weights) HIP_TRY(hipMemcpy(debug_out->edge_weights, a.weights, (size_t)Ec * 8, hipMemcpyDeviceToHost));
            if (debug_out->edge_weights_int) HIP_TRY(hipMemcpy(debug_out->edge_weights_int, a.weights_i, (size_t)Ec * 4, hipMemcpyDeviceToHost));
        }
    }
    if (edges_found) *edges_found = E;
    if (hmisc[2] & 2) {
        set_error("segment: more graph edges than the edge table holds");
        return -2;
    }
    if (use_graphcut && (hmisc[2] & 1)) {
        set_error("cut_general_graph: smoothness term is larger than GCO_MAX_ENERGYTERM");
        return -1;
    }
    if (use_graphcut && hmisc[3] != 0) {
        set_error("alpha_expansion: max-flow did not converge");
        return -1;
    }
    return 0;
}


// the whole colour pipeline of one image in ONE call: a worker thread of the Python layer spends a step here, outside
// the interpreter lock (H2D, SLIC with one host synchronisation for the label count, features, fused back half, D2H)
int imsegm_image2d_run_color(imsegm_image2d *im, const void *host_pixels, int dtype, int minmax_normalize, int n_segments,
                             double compactness, const double *taps, int radius, int max_iter, int start_label, int slic_zero,
                             int feature_mask, const imsegm_gmm *gmm, int n_classes, const double *pairwise, int edge_type,
                             double edge_cost, int use_graphcut, const int32_t *classes_lut, int32_t *segm_out, double *soft_out,
                             int *n_labels_out)
{
    if (imsegm_image2d_upload(im, host_pixels, dtype)) return -1;
    int n_labels = 0;
    if (imsegm_image2d_slic(im, minmax_normalize, n_segments, compactness, taps, radius, taps, radius, taps, radius, max_iter, 1, 0.5,
                            3.0, start_label, 0, slic_zero, &n_labels))
        return -1;
    if (n_labels_out) *n_labels_out = n_labels;
    if (imsegm_image2d_features_color(im, feature_mask, nullptr)) return -1;
    return imsegm_image2d_segment(im, gmm, nullptr, n_classes, pairwise, edge_type, edge_cost, use_graphcut, classes_lut, segm_out,
                                  soft_out, nullptr, nullptr, nullptr);
}



}  // extern "C"
