// slic.h -- internal (C++) interface between the C-ABI layer (api.hip) and the kernel files.
#pragma once
#include "common.h"
#include <algorithm>
#include <cfloat>
#include <cmath>

namespace imsegm {

enum { DT_U8 = 0, DT_F64 = 1, DT_F32 = 2 };

// half of a symmetric correlation kernel, w[0] = centre tap; r < 0: axis not filtered
struct Taps {
    int r;
    double w[17];
};

// tile geometry of the assignment kernel and the per-tile candidate records (64 bytes each)
constexpr int SLIC_TILE_X = 64, SLIC_TILE_Y = 32, SLIC_MAXC = 64;
struct Cand {
    double cy, cx, cL, ca, cb;   // exact (fp64) centroid
    int4 win;                    // search window {ymin, ymax, xmin, xmax}
    int k;                       // centroid index
    float ry, rx;                // fp32 position relative to the tile origin
    float fL, fa, fb;            // fp32 colour
    double mdc;                  // SLICO: largest colour distance seen inside the segment (1 otherwise)
};

// compact fp32 record of the same candidate for the pre-selection loop of k_slic_assign_dot (32 bytes, read
// with one scalar load).  Coordinates are relative to the centre (row 16, column 32) of the 64 x 32 bin
// tile, colours relative to the tile's reference colour; the distance of pixel (Y, X, f) is, up to a term
// that does not depend on the candidate,  q0 + qy*Y + qx*X + qL*fL + qa*fa + qb*fb.
struct Rec32 {
    float q0, qx, qy, qL, qa, qb;
    float lbt;                   // lower bound of the distance over the whole tile (sort key, ascending)
    uint32_t meta;               // window relative to the tile: rlo | rhi << 8 | xlo << 16 | xhi << 24
};
// per-tile header of the record table
struct TileInfo {
    double ref[3];               // reference colour (exact) subtracted from pixels and centroids
    float Qy, Qx, QL, Qa, Qb;    // max |qy|, |qx|, |qL|, |qa|, |qb| over the candidates of the tile
    int count;                   // same as tile_count
};

// device-side SLIC state for one 2-D image (all pointers are device pointers)
constexpr int SLIC_DRIFT_SLOTS = 64;
struct SlicState {
    int H, W, K;
    int step_y, step_x;
    double spatial_weight;          // 1 / step^2  (_slic.pyx "invxywt")
    double *cy, *cx, *cL, *ca, *cb; // centroid table, SoA fp64 [K]
    int4 *win;                      // integer search window {ymin, ymax, xmin, xmax} [K]
    long long *acc;                 // [K][9] = n, sum y, sum x, (hi, lo) limbs of the fixed-point sums of L, a, b
    const double *premax;           // [1] max |pre-processed value| (device): fixes the fixed-point format (common.h)
    Cand *tile_cands;               // [n_tiles][SLIC_MAXC] nearest-first candidate centroids per tile
    int *tile_count;                // [n_tiles] list length (negative: more than the list holds)
    Rec32 *tile_rec;                // [n_tiles][SLIC_MAXC] fp32 records, same slot order as tile_cands
    TileInfo *tile_info;            // [n_tiles]
    int *tile_k;                    // [n_tiles][SLIC_MAXC] centroid index per slot (coalesced copy of Cand::k)
    uint32_t *tile_rows;            // [n_tiles][SLIC_MAXC] bit y = row y of the 64 x 32 tile lies inside the slot's search window
    int *leftover;                  // [N] pixels to be accumulated by k_slic_leftover
    int *leftover_count;            // [1]
    int fast32;                     // 1: fp32 pre-selection allowed (Lab bounded by lab_bound)
    float kappa;                    // relative decision margin of the fp32 pass (see k_slic_assign)
    int grid_y0, grid_dy, grid_x0, grid_dx, grid_nx;   // initial centroid grid (skimage regular_grid)
    long long *phase_prof;          // profiling aid (env IMSEGM_PHASE_PROF): per-phase cycle sums, or null
    int debug;                      // profiling aid (env IMSEGM_DEBUG_ASSIGN): ablation bits, results invalid
    int assign_units;               // 64 x 4 units per wave of k_slic_assign_dot: 2 (a workgroup = one bin tile) or 1
    int slico;                      // skimage slic_zero: colour distance / mdc[k], mdc updated after every sweep
    double *mdc;                    // [K] max_dist_color of _slic.pyx (starts at 1)
    int *drift;                     // [SLIC_DRIFT_SLOTS] per sweep: max displacement of a centroid from its grid node
    // centroid update inside the assignment kernel (k_slic_assign_dot): the workgroup that adds the LAST contribution to a
    // centroid -- arrivals are counted per centroid against the number of workgroups its search window meets -- divides the
    // sums itself; no k_centroid_finalize launch between the sweeps
    int *done;                      // [K] arrivals of the current sweep (null: separate finalize launches)
    int *fail_host;                 // page-locked word raised when a contribution bypassed the arrival count (the host redoes the image)
    int fuse_finalize;              // this launch updates the centroids itself
    int drift_slot_next;            // drift slot the updated centroids report into
    size_t zs;                      // several images per launch (ZBatch): bytes between the buffers of consecutive images
};

int launch_minmax(const void *src, int dtype, size_t n, unsigned long long *keys, double *out2, hipStream_t st, double *also_zero = nullptr,
                  ZBatch zb = ZBatch());
// premax_dev[0] receives max |value| of the result planes
int launch_preprocess_color2d(const void *img, int dtype, int H, int W, int normalize, const double *minmax_dev,
                              const Taps &tz, const Taps &ty, const Taps &tx, double ratio, double *bufA,
                              double *bufB, double *premax_dev, hipStream_t st, bool premax_zeroed = false, ZBatch zb = ZBatch());
int launch_absmax_f64(const double *src, size_t n, double *out_dev, hipStream_t st);
// optional HIP-event hooks around the dominant kernel (api.hip profiler)
struct ProfHook {
    void *user = nullptr;
    int (*begin)(void *, int) = nullptr;
    void (*end)(void *, int) = nullptr;
    // event pair to attach to ONE kernel launch (hipExtLaunchKernelGGL): the pair then carries the begin / end
    // timestamps of that dispatch itself, without the gap an hipEventRecord in front of the launch adds
    void (*pair)(void *, int, hipEvent_t *, hipEvent_t *) = nullptr;
};
int slic_prepare_device();
// diagnostics: images whose sweeps were redone with separate finalize launches (the centroid update inside the assignment kernel
// handed them back); persistent_runs: always 0 since round 6
void slic_sweep_counters(long *persistent_runs, long *fallback_runs);
void slic_sweep_note_fallback();
// *fused_update: the centroid update ran inside the assignment kernel -- the caller reads s.fail_host after its next synchronisation
int launch_slic_iterations(SlicState s, const double *lab, const double *init_yx_dev, int32_t *labels, int max_iter,
                           int max_cand, const ProfHook &prof, hipStream_t st, bool *fused_update = nullptr, ZBatch zb = ZBatch());

// volume.hip -------------------------------------------------------------------------------------
struct VolState {
    int D, H, W, K;
    int step_z, step_y, step_x;
    double spatial_weight;          // 1 / step^2
    double sz, sy, sx;              // voxel spacing
    double *cen;                    // [K][4] = cz, cy, cx, value
    int *win;                       // [K][6] = zmin, zmax, ymin, ymax, xmin, xmax
    long long *acc;                 // [K][6] = n, sum z, sum y, sum x, (hi, lo) limbs of the fixed-point value sum
    const double *premax;           // [1] max |pre-processed value| (device)
    int grid_0[3], grid_d[3], grid_n[3];
    // candidate lists per brick of VOL_BZ x VOL_BY x VOL_BX voxels, filled by the centroids themselves
    // (k_vol_scatter): a wave of the assignment kernel only looks at the list of its brick
    int nbz, nby, nbx;              // bricks per axis
    int brick_cap;                  // entries per brick list
    int *brick_count;               // [n_bricks] (may exceed brick_cap: that brick scans the whole table)
    int *brick_list;                // [n_bricks][brick_cap]
    int *brick_entries;             // float32 volumes: [n_bricks][brick_cap][12] -- a VolEntry per list entry (volume.hip), brick_list unused
    // float32 volumes (scikit-image keeps them in float32): float32 centroid table and the bounding boxes of the
    // segments' voxels that the order-preserving update walks
    float *cen32;                   // [K][4] = cz, cy, cx, value
    int *bbox;                      // [K][6] = zmin, zmax, ymin, ymax, xmin, xmax (inclusive; zmax < zmin: no voxel)
};
constexpr int VOL_BX = 64, VOL_BY = 16, VOL_BZ = 16;
int launch_vol_preprocess(const void *src, int dtype, double off, double scale, int D, int H, int W, const Taps &tz, const Taps &ty,
                          const Taps &tx, double ratio, double *bufA, double *bufB, hipStream_t st);
int launch_vol_slic(VolState s, const double *vol, int32_t *labels, int max_iter, hipStream_t st);
// float32 volume: result of the pre-processing is a float32 plane in bufB
int launch_vol_preprocess_f32(const float *src, int D, int H, int W, const Taps &tz, const Taps &ty, const Taps &tx, double ratio,
                              double *bufA, double *bufB, hipStream_t st);
int launch_vol_slic_f32(VolState s, const float *vol, int32_t *labels, int max_iter, hipStream_t st, const ProfHook *prof = nullptr);
int launch_label_cc(int32_t *labels_inout, int D, int H, int W, int32_t *parent, int32_t *newlabel, int32_t *blocksum,
                    int32_t *total_dev, hipStream_t st);
int launch_vol_adjacency(const int32_t *labels, int D, int H, int W, int K, int words, uint32_t *bitmap, long long *cacc,
                         double *centres, uint8_t *present, hipStream_t st);
int launch_vol_adjacency_table(const int32_t *labels, int D, int H, int W, int K, int32_t *table, int cap, int *overflow, long long *cacc,
                               double *centres, uint8_t *present, hipStream_t st);
int launch_edge_extract_table(const int32_t *table, int K, int cap, int32_t *rowcount, int32_t *edges_out, int edge_capacity,
                              int32_t *n_edges_dev, hipStream_t st);
int launch_edge_extract(const uint32_t *bitmap, int K, int words, int32_t *rowcount, int32_t *edges_out, int edge_capacity,
                        int32_t *n_edges_dev, hipStream_t st);

// connectivity.hip ------------------------------------------------------------------------------
struct ConnWork {
    int32_t *parent;      // [N] union-find forest / component id (root = min raster index)
    int32_t *csize;       // [N] component size, valid at roots
    int32_t *newlabel;    // [N] final label per root
    int32_t *adjptr;      // [N] small component -> root of the component it merges into (-1: none)
    int32_t *queue;       // [N] BFS queues
    uint8_t *visited;     // [N]
    int32_t *blocksum;    // [nblocks + 1]
    int32_t *list;        // [N] compacted list of component roots
    int32_t *counters;    // [16] misc device counters
    int32_t *slotmap;     // [N] root -> index in the small-component list
    int32_t *bbox;        // [N] bounding boxes of small components (N/12 x 6) + fallback list (N/2)
    int32_t *dense;       // lists of the 2-D tile path: CONN_DENSE_INTS (kept roots, hand-overs) + per-tile slots of local roots
    size_t dense_ints;    // ints available behind `dense` (the tile path checks that its lists fit before it runs)
};
constexpr size_t CONN_DENSE_INTS = 2 * 65536 + 16 * 4096;
constexpr int CONN_TILE_PIXELS = 64 * 32, CONN_TILE_SLOTS = 256;
// bytes of the int32 scratch behind a ConnWork for n pixels (2-D: H x W, tiles of 64 x 32 with 64 slots x 3 lists + a count)
static inline size_t conn_i32_bytes(size_t n, size_t H = 0, size_t W = 0)
{
    const size_t n_tiles = ((W + 63) / 64) * ((H + 31) / 32);
    return n * 4 * 8 + ((n / 4096) + 64) * 4 + 256 + (CONN_DENSE_INTS + n_tiles * (3 * CONN_TILE_SLOTS + 1)) * 4;
}
// measure.label of a map whose labels > 0 are connected sets each (what launch_enforce_connectivity writes): renumbering by first
// voxel; *total_dev = the number of labels > 0
int launch_label_connected(int32_t *labels_inout, size_t n, int n_labels, int32_t *first, uint32_t *bitmap, int32_t *blocksum,
                           int32_t *counters, int32_t *total_dev, hipStream_t st);
long conn_general_runs();      // diagnostic: 2-D maps that left the tile path so far
long gc_grid_fallbacks();      // diagnostic: grid-wide cuts given up (a workgroup not resident) and redone by the single workgroup
int launch_enforce_connectivity(const int32_t *labels_in, int D, int H, int W, long min_size, long max_size,
                                int start_label, ConnWork w, int32_t *labels_out, int *n_labels_out_host,
                                hipStream_t st);
// the 2-D tile path on zb.nz maps of one size in one chain of launches and ONE host synchronisation; n_labels_out[b] < 0: image b
// must go through launch_enforce_connectivity alone (the general path)
int launch_enforce_connectivity_batch(const int32_t *labels_in, int H, int W, long min_size, long max_size, int start_label, ConnWork w,
                                      int32_t *labels_out, int *n_labels_out, hipStream_t st, ZBatch zb, int32_t *stage_dev);
                                      // (stage_dev: zb.nz * 16 ints of device memory outside the images' slices)

// stats.hip ---------------------------------------------------------------------------------------
// hist[K][nb] (zeroed here) += 1 per pixel with label k in [0, K) and annotation a in [0, nb)  (labeling.py:208-247)
int launch_label_hist(const int32_t *labels, const int32_t *annot, size_t n, int K, int nb, unsigned long long *hist,
                      hipStream_t st);
// per-superpixel colour statistics (features_cython.pyx:59-141): sums of v, v*v (float32 product)
// and (v - mean32)^2 (float32), exact fixed-point accumulation. acc: [K][13] int64 scratch.
int launch_color_stats(const void *img, int dtype, const int32_t *labels, int H, int W, int K, double maxabs,
                       int want_var, long long *acc, double *mean_out, double *energy_out, double *var_out,
                       float *mean32_scratch, hipStream_t st, int planar = 0, int prescale = 0, double mul = 1.0,
                       double div = 1.0, long plane_stride = -1, const double *ssq_dev = nullptr, ZBatch zb = ZBatch());

// texture.hip -------------------------------------------------------------------------------------
// the separable kernels of one battery: `groups` kernels of `rank` components (per component the x taps then the y taps of the
// flipped kernel); merge: the maximum with what the battery's dense kernels have already written to resp
struct SepJob {
    double *resp;
    const double *taps;
    int groups, rank, merge;
};
constexpr int SEP_MAX_JOBS = 5;
struct SepJobs {
    int n;
    SepJob job[SEP_MAX_JOBS];
};
int launch_battery_dense(const double *planes, int H, int W, const double *wgt_dev, int nk, int radius, double clip, double *resp,
                         hipStream_t st, int P, int parity);
size_t sep_sumsq_scratch(int H, int W, int P, int radius, int jobs);
int launch_battery_sep(const double *planes, int H, int W, int radius, double clip, const SepJobs &jobs, hipStream_t st, int P,
                       double *ssq_scratch, double *const *ssq_out);
int launch_response_sumsq(const double *resp, size_t count, double *partial, double *sumsq_dev, hipStream_t st);
// fullpad: device scratch of 2 * radius + 1 + 16 doubles (the zero-padded full tap table the column pass reads)
int launch_texture_prepare(const void *img, int dtype, int H, int W, const double *taps_dev, int radius,
                           const double *mix_dev, double *planes, double *tmpA, double *tmpB, hipStream_t st, double *fullpad);
int launch_texture_prepare_volume(const void *vol, int dtype, int P, int H, int W, const double *taps_dev, int radius, double *planes,
                                  double *tmpA, double *tmpB, hipStream_t st, double *fullpad);
// P planes of H x W (3 colour channels, or the D slices of a gray volume)
int launch_filter_battery(const double *planes, int H, int W, const double *wgt_dev, int nk, int radius, double clip,
                          double *resp, double *partial, double *sumsq_dev, hipStream_t st, int P = 3, const double *sep_dev = nullptr,
                          int sep_groups = 0, int sep_rank = 0, int parity = 0);
// (nk dense kernels -- 0, 1, 2, 4, 6 or 8 -- plus sep_groups separable kernels of sep_rank components each: per component the x
// taps then the y taps of the flipped kernel, 2 * (2 radius + 1) doubles; the response is the maximum over all of them;
// parity +1 / -1: every dense kernel is even / odd under the point reflection, bit for bit -- the caller has checked)
// the caller reserves S * S * nk + S * (S + 16) * nk doubles behind wgt_dev (the weights, then their row-padded copy)

// graph.hip ---------------------------------------------------------------------------------------
int launch_adjacency_centres(const int32_t *labels, int H, int W, int K, uint32_t *bitmap, long long *cacc,
                             int32_t *edges_out, int edge_capacity, int32_t *n_edges_dev, double *centres_out,
                             uint8_t *present_out, int32_t *rowcount, hipStream_t st);
// bitmap (row b, column a < b) + centre sums only; launch_adjacency_centres = this + launch_edge_extract
int launch_adjacency_bitmap(const int32_t *labels, int H, int W, int K, uint32_t *bitmap, long long *cacc, double *centres_out,
                            uint8_t *present_out, hipStream_t st, ZBatch zb = ZBatch());
int launch_gather_labels(const int32_t *lut, const int32_t *idx, size_t n, int32_t *out, hipStream_t st, ZBatch zb = ZBatch());
// zero `bytes` bytes of every image of a batch (one image: hipMemsetAsync)
int launch_zero(void *ptr, size_t bytes, hipStream_t st, ZBatch zb);
// `rows` rows of `bytes` bytes (multiples of 4) from src + r * src_stride to dst + r * dst_stride
int launch_copy_rows(void *dst, size_t dst_stride, const void *src, size_t src_stride, size_t bytes, int rows, hipStream_t st);
int launch_gather_proba(const double *lut, int C, const int32_t *idx, size_t n, double *out, hipStream_t st);

// output.hip -------------------------------------------------------------------------------------
// rects: four border strips {r0, r1, c0, c1} (rows [r0, r1), columns [c0, c1)); a pixel counts once per strip that holds it
int launch_boundary_minmax(const int32_t *labels, int W, const int rects[16], int32_t *out2_dev, hipStream_t st);
int launch_boundary_hist(const int32_t *labels, int W, const int rects[16], unsigned long long *hist_dev, int nb, hipStream_t st);
int launch_swap_labels(int32_t *labels, size_t n, int a, int b, hipStream_t st);
int launch_narrow_labels_u8(const int32_t *src, uint8_t *dst, size_t n, hipStream_t st);
int launch_narrow_soft_f32(const double *src, float *dst, size_t n, hipStream_t st);
int launch_count_nonfinite(const void *src, int dtype, size_t n, unsigned int *count_dev, hipStream_t st);

// median.hip -------------------------------------------------------------------------------------
int launch_gradient_image(const void *src, void *dst, int dtype, int S, int H, int W, int C, hipStream_t st);
size_t median_scratch_bytes(size_t n, int K);
int launch_segment_median(const void *img, int dtype, int C, size_t n, const int32_t *labels, int K, void *scratch, size_t scratch_bytes,
                          double *out, hipStream_t st);

// natives.hip ------------------------------------------------------------------------------------
int launch_label_hist2d(const int16_t *segm, int H, int W, const int32_t *windows, int P, const int16_t *selem, int SH, int SW,
                        int nb_labels, unsigned int *hist, hipStream_t st);
int launch_ray_features_binary2d(const int8_t *seg, int H, int W, const int32_t *positions, int P, const float *grad, int A, int edge,
                                 float *out, hipStream_t st);

// graphcut.hip ------------------------------------------------------------------------------------
struct GcProblem {
    int K, C, E;            // E: number of edges, or their capacity when E_dev is given
    const int32_t *E_dev = nullptr;   // number of edges on the device (fused pipeline: no host round trip)
    const int32_t *K_dev = nullptr;   // number of sites on the device, K its upper bound (a batch: the graphs differ in size)
    const int32_t *edges;   // [E][2], a < b
    const int32_t *w;       // [E]
    const int32_t *unary;   // [K][C]
    const int32_t *smooth;  // [C][C]
    // the integer smoothness matrix is a metric (smooth_is_metric): every expansion move is then solved exactly, and a move that
    // repeats the label of the last accepted move cannot lower the energy -- the kernel answers it without a max-flow
    int metric = 0;
};
// V(a, a) = 0, V(a, b) = V(b, a) >= 0, V(a, b) <= V(a, c) + V(c, b) on the integers GCO works with
static inline int smooth_is_metric(const int32_t *s, int C)
{
    for (int a = 0; a < C; ++a) {
        if (s[a * C + a] != 0) return 0;
        for (int b = 0; b < C; ++b)
            if (s[a * C + b] < 0 || s[a * C + b] != s[b * C + a]) return 0;
    }
    for (int a = 0; a < C; ++a)
        for (int b = 0; b < C; ++b)
            for (int c = 0; c < C; ++c)
                if ((long long)s[a * C + b] > (long long)s[a * C + c] + (long long)s[c * C + b]) return 0;
    return 1;
}
int launch_alpha_expansion(GcProblem p, const int32_t *arc_start, const int32_t *arc_to, const int32_t *arc_rev,
                           const int32_t *edge_arc, int n_iter, int32_t *labels_dev, long long *energy_dev,
                           int32_t *status_dev, void *work, hipStream_t st, ZBatch zb = ZBatch());
size_t alpha_expansion_work_bytes(int K, int E);

// terms.hip ---------------------------------------------------------------------------------------
struct TermsArgs {
    const int *Kp;                 // number of superpixels (device)
    int K_cap;                     // upper bound of it known to the host (launch geometry)
    const int *Ep;                 // number of edges (device)
    int edge_capacity;
    int F, C;
    const double *features;        // [K][F]
    // class model (gmm != 0): StandardScaler + full-covariance Gaussian mixture
    int gmm;
    const double *scaler_mean, *scaler_scale;   // [F] or null
    const double *prec_chol;       // [C][F][F]
    const double *mu_proj;         // [C][F] = means @ prec_chol (host, as scikit-learn forms it)
    const double *log_det;         // [C]
    const double *log_w;           // [C]
    double const_term;             // F * log(2 pi)
    double *proba;                 // [K][C] in (gmm == 0) / out
    // edges
    int edge_type;                 // 0 const, 1 spatial, 2 model lT, 3 model l1, 4 model l2, 5 features
    int spatial_norm;              // divide by the relative centre distance (graph_cuts.py:647: only 'model', 'features', 'spatial')
    double edge_cost;
    const int32_t *edges;          // [E][2]
    const double *centres;         // [K][ndim]
    int ndim;
    double *edge_dist, *edge_len;  // [E] scratch
    // outputs
    double *unary;                 // [K][C]
    double *weights;               // [E]
    const double *pairwise;        // [C][C]
    double pairwise_max;
    int32_t *unary_i, *weights_i;  // pyGCO integers
    int smooth_max;                // max |int(pairwise * 100)|
    int32_t *status;               // bit 0: smoothness term above GCO_MAX_ENERGYTERM, bit 1: edge list overflow
    double *scalars;               // [8] debug: mean len, mean dist, std dist, umax, wmax, dwf
    double *fstd;                  // [2][F] scratch (edge type 'features')
    size_t zs;                     // several images per launch (ZBatch): bytes between the buffers of consecutive images
};
int launch_features_assemble(const double *mean, const double *energy, const double *var, int K, int mask, double *out,
                             hipStream_t st, int row_stride = 0, int col0 = 0, ZBatch zb = ZBatch());
// symmetric bitmap -> edge list ordered by (b, a), CSR arcs in ascending neighbour order, reverse arcs, edge -> arc table
int launch_graph_csr(uint32_t *bitmap, const int *K_dev, int K_cap, int words, int32_t *wordprefix, int32_t *deg, int32_t *deg_low,
                     int32_t *arc_start, int32_t *edge_start, int32_t *n_edges_dev, int edge_capacity, int32_t *edges,
                     int32_t *arc_to, int32_t *arc_rev, int32_t *edge_arc, hipStream_t st, ZBatch zb = ZBatch());
// the same out of the symmetric neighbour table of a label volume (rows of cap <= 64 slots; the rows come back sorted)
// (*overflow: a row was too narrow -- raised by the adjacency kernel; the graph then comes out without edges)
int launch_graph_csr_table(int32_t *table, const int *K_dev, int K_cap, int cap, const int *overflow, int32_t *deg, int32_t *deg_low,
                           int32_t *arc_start, int32_t *edge_start, int32_t *n_edges_dev, int edge_capacity, int32_t *edges,
                           int32_t *arc_to, int32_t *arc_rev, int32_t *edge_arc, hipStream_t st);
int launch_gc_terms(const TermsArgs &a, hipStream_t st, int nz = 1);          // (a.zs: the stride of a batch)
int launch_unary_argmin(const double *unary, const int *K_dev, int K_cap, int C, int32_t *labels, hipStream_t st, ZBatch zb = ZBatch());
int launch_label_lut(const int32_t *graph_labels, const int *K_dev, int K_cap, const int32_t *classes, int32_t *lut, hipStream_t st,
                     ZBatch zb = ZBatch());

}  // namespace imsegm
