// api.hip -- C ABI (include/imsegm_hip.h) and host-side orchestration of libimsegm_hip.so.
#include "../../include/imsegm_hip.h"
#include "slic.h"

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace imsegm {

static thread_local std::string g_error;

void set_error(const std::string &msg) { g_error = msg; }

bool hip_ok(hipError_t e, const char *what, const char *file, int line)
{
    if (e == hipSuccess) return true;
    char buf[512];
    snprintf(buf, sizeof(buf), "HIP error %d (%s) in `%s` at %s:%d", (int)e, hipGetErrorString(e), what, file, line);
    g_error = buf;
    return false;
}

static Knobs read_knobs()
{
    auto flag = [](const char *n) { return getenv(n) != nullptr; };
    auto num = [](const char *n, int dflt) { const char *e = getenv(n); return e ? atoi(e) : dflt; };
    Knobs k;
    k.slic_graph = flag("IMSEGM_SLIC_GRAPH");
    k.slic_persistent = flag("IMSEGM_SLIC_PERSISTENT");
    k.pre_3pass = flag("IMSEGM_PRE_3PASS");
    k.separate_finalize = flag("IMSEGM_SEPARATE_FINALIZE");
    k.fuse_finalize = flag("IMSEGM_FUSE_FINALIZE");
    k.sweeps_force_fail = flag("IMSEGM_SWEEPS_FORCE_FAIL");
    k.conn_general = flag("IMSEGM_CONN_GENERAL");
    k.gc_no_topo_regs = flag("IMSEGM_GC_NO_TOPO_REGS");
    k.sep_wide_tile = flag("IMSEGM_SEP_WIDE_TILE");
    k.vol_update_wave = flag("IMSEGM_VOL_UPDATE_WAVE");
    k.cc_merge_full = flag("IMSEGM_CC_MERGE_FULL");
    k.adjacency_table = flag("IMSEGM_ADJACENCY_TABLE");
    k.brick_cap = num("IMSEGM_BRICK_CAP", 0);
    k.gc_lds_level = num("IMSEGM_GC_LDS_LEVEL", 4);
    k.gc_threads = num("IMSEGM_GC_THREADS", 0);
    k.gc_grid_min_sites = num("IMSEGM_GC_GRID_MIN_SITES", 0);
    k.gc_grid_blocks = num("IMSEGM_GC_GRID_BLOCKS", 0);
    k.gc_one_workgroup = flag("IMSEGM_GC_ONE_WORKGROUP");
    k.gc_grid_test_absent = flag("IMSEGM_GC_GRID_TEST_ABSENT");
    k.fused_bitmap_mb = num("IMSEGM_FUSED_BITMAP_MB", 0);
    k.sweeps_blocks_per_cu = num("IMSEGM_SWEEPS_BLOCKS_PER_CU", 0);
    k.sweeps_per_launch = num("IMSEGM_SWEEPS_PER_LAUNCH", 0);
    const char *d = getenv("IMSEGM_PHASE_DUMP");
    k.phase_dump = d ? d : "";
    return k;
}
// (a snapshot is never freed: a thread may still hold a reference to the one it started its call with)
static std::atomic<const Knobs *> g_knobs{ nullptr };
const Knobs &knobs()
{
    const Knobs *k = g_knobs.load(std::memory_order_acquire);
    if (!k) {
        const Knobs *fresh = new Knobs(read_knobs());
        if (g_knobs.compare_exchange_strong(k, fresh, std::memory_order_acq_rel)) k = fresh;
        else delete fresh;
    }
    return *k;
}
void reload_knobs() { g_knobs.store(new Knobs(read_knobs()), std::memory_order_release); }

}  // namespace imsegm

#include "session.h"

std::atomic<bool> g_runtime_started{ false };

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

int imsegm_host_alloc(size_t bytes, void **ptr_out)
{
    g_runtime_started.store(true);
    if (!ptr_out) {
        set_error("null argument");
        return -1;
    }
    HIP_TRY(hipHostMalloc(ptr_out, bytes ? bytes : 1, hipHostMallocDefault));
    return 0;
}

void imsegm_host_free(void *ptr)
{
    if (ptr) (void)hipHostFree(ptr);
}

int imsegm_device_alloc(int device, size_t bytes, void **ptr_out)
{
    g_runtime_started.store(true);
    if (!ptr_out) {
        set_error("null argument");
        return -1;
    }
    HIP_TRY(hipSetDevice(device));
    HIP_TRY(hipMalloc(ptr_out, bytes ? bytes : 1));
    return 0;
}

void imsegm_device_free(void *ptr)
{
    if (ptr) (void)hipFree(ptr);
}

int imsegm_set_device(int device)
{
    g_runtime_started.store(true);
    HIP_TRY(hipSetDevice(device));
    return 0;
}

int imsegm_ctx_stream(imsegm_ctx *ctx, void **stream_out)
{
    if (bind(ctx)) return -1;
    *stream_out = ctx->stream;
    return 0;
}

int imsegm_ctx_copy(imsegm_ctx *ctx, void *dst, const void *src, size_t bytes, int synchronize)
{
    if (bind(ctx)) return -1;
    HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDefault, ctx->stream));
    if (synchronize) HIP_TRY(hipStreamSynchronize(ctx->stream));
    return 0;
}

const char *imsegm_last_error(void) { return g_error.c_str(); }
int imsegm_version(void) { return 100; }

int imsegm_init(int hardware_queues)
{
    if (hardware_queues < 1 || hardware_queues > 64) {
        set_error("imsegm_init: 1..64 hardware queues");
        return -1;
    }
    if (g_runtime_started.load() || getenv("GPU_MAX_HW_QUEUES")) return 1;
    char buf[16];
    snprintf(buf, sizeof(buf), "%d", hardware_queues);
    return setenv("GPU_MAX_HW_QUEUES", buf, 0) == 0 ? 0 : -1;
}

int imsegm_device_pci_bus_id(int device, char *id_out, int capacity)
{
    g_runtime_started.store(true);
    if (!id_out || capacity < 16) {
        set_error("device_pci_bus_id: a buffer of at least 16 characters");
        return -1;
    }
    HIP_TRY(hipDeviceGetPCIBusId(id_out, capacity, device));
    return 0;
}

int imsegm_device_mem_info(int device, size_t *free_bytes_out, size_t *total_bytes_out)
{
    if (!free_bytes_out || !total_bytes_out) {
        set_error("mem_info: null output");
        return -1;
    }
    g_runtime_started.store(true);
    HIP_TRY(hipSetDevice(device));
    HIP_TRY(hipMemGetInfo(free_bytes_out, total_bytes_out));
    return 0;
}

int imsegm_device_count(int *count_out)
{
    g_runtime_started.store(true);
    int c = 0;
    hipError_t e = hipGetDeviceCount(&c);
    if (e != hipSuccess) {
        c = 0;
        (void)hipGetLastError();
    }
    *count_out = c;
    return 0;
}

int imsegm_ctx_create(int device, imsegm_ctx **ctx_out)
{
    g_runtime_started.store(true);
    int c = 0;
    HIP_TRY(hipGetDeviceCount(&c));
    if (device < 0 || device >= c) {
        set_error("no such HIP device");
        return -1;
    }
    HIP_TRY(hipSetDevice(device));
    imsegm_ctx *ctx = new imsegm_ctx();
    ctx->device = device;
    HIP_TRY(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
    *ctx_out = ctx;
    return 0;
}

void imsegm_ctx_destroy(imsegm_ctx *ctx)
{
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    ctx->collect();
    for (auto e : ctx->pool) (void)hipEventDestroy(e);
    ctx->gc_buf.release();
    ctx->aux_buf.release();
    if (ctx->pinned_ev) (void)hipEventDestroy(ctx->pinned_ev);
    if (ctx->pinned) (void)hipHostFree(ctx->pinned);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

int imsegm_ctx_synchronize(imsegm_ctx *ctx)
{
    if (bind(ctx)) return -1;
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return 0;
}

int imsegm_ctx_profile_enable(imsegm_ctx *ctx, int enable)
{
    if (bind(ctx)) return -1;
    ctx->collect();
    ctx->profile = enable != 0;
    return 0;
}

int imsegm_ctx_profile_reset(imsegm_ctx *ctx)
{
    if (bind(ctx)) return -1;
    ctx->collect();
    for (int i = 0; i < PG_COUNT; ++i) {
        ctx->acc_ms[i] = 0;
        ctx->acc_n[i] = 0;
    }
    return 0;
}

int imsegm_ctx_profile_get(imsegm_ctx *ctx, int group, double *total_ms_out, int *count_out)
{
    if (bind(ctx)) return -1;
    if (group < 0 || group >= PG_COUNT) {
        set_error("bad profile group");
        return -1;
    }
    ctx->collect();
    *total_ms_out = ctx->acc_ms[group];
    *count_out = ctx->acc_n[group];
    return 0;
}

int imsegm_image2d_create(imsegm_ctx *ctx, int height, int width, imsegm_image2d **img_out)
{
    if (bind(ctx)) return -1;
    if (height <= 0 || width <= 0 || (long)height * width > 0x3fffffffL) {
        set_error("bad image size");
        return -1;
    }
    imsegm_image2d *im = new imsegm_image2d();
    im->ctx = ctx;
    im->H = height;
    im->W = width;
    im->n = (size_t)height * width;
    *img_out = im;
    return 0;
}

void imsegm_image2d_destroy(imsegm_image2d *im)
{
    if (!im) return;
    (void)hipSetDevice(im->ctx->device);
    (void)hipStreamSynchronize(im->ctx->stream);
    DevBuf *all[] = { &im->img, &im->labA, &im->labB, &im->nearest, &im->labels, &im->conn_i32, &im->conn_u8, &im->small,
                      &im->cent, &im->tiles, &im->feat, &im->graph, &im->gather_lut, &im->gather_out_i, &im->gather_out_f,
                      &im->tex_planes, &im->tex_resp, &im->tex_small, &im->vol_cent, &im->annot, &im->hist, &im->featK, &im->seg,
                      &im->gseg, &im->sweeps, &im->narrow };
    for (auto b : all) b->release();
    if (im->slic_fail_host) (void)hipHostFree(im->slic_fail_host);
    if (im->slic_exec) (void)hipGraphExecDestroy(im->slic_exec);
    delete im;
}

int imsegm_image2d_upload(imsegm_image2d *im, const void *host_pixels, int dtype)
{
    if (!im || bind(im->ctx)) return -1;
    if (wrong_kind(im, false)) return -1;
    im->tex_ready = false;
    size_t es = dtype == IMSEGM_U8 ? 1 : dtype == IMSEGM_F32 ? 4 : dtype == IMSEGM_F64 ? 8 : 0;
    if (!es) {
        set_error("unsupported dtype");
        return -1;
    }
    size_t bytes = im->n * 3 * es;
    if (im->img.ensure(bytes + 16)) return -1;
    HIP_TRY(hipMemcpyAsync(im->img.p, host_pixels, bytes, hipMemcpyHostToDevice, im->ctx->stream));
    // a page-locked source (imsegm_host_alloc) is read by the DMA engine when the stream gets there: no wait here,
    // the caller keeps the buffer untouched until the next call that synchronises (slic does)
    if (!is_pinned(host_pixels)) HIP_TRY(hipStreamSynchronize(im->ctx->stream));
    im->dtype = dtype;
    im->feat_mask = 0;
    im->place_F = 0;
    return 0;
}

static ConnWork make_conn_work(imsegm_image2d *im);

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
    bool used_persistent = false;
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
        // scratch of the persistent sweep kernel (all sweeps after the first in one launch); its failure word is page-locked
        // host memory that is read after the next synchronisation of this call (the connectivity stage ends with one)
        void *sweep_scratch = nullptr;
        if (knobs().slic_persistent) {
            if (im->sweeps.ensure(sweep_work_bytes(K, max_iter, (int)n_tiles, cdiv(H, SLIC_TILE_Y)))) return -1;
            sweep_scratch = im->sweeps.p;
        }
        if (launch_slic_iterations(s, im->labA.as<double>(), init_dev, im->nearest.as<int32_t>(), max_iter, max_candidates, hook, st,
                                   sweep_scratch, im->slic_fail_host, &used_persistent))
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
            // the persistent kernel gave the image back (more candidates in a tile than a list holds, a centroid far from its
            // grid node, an uncovered pixel): the sweeps again, one launch each -- they take every case
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
        if (used_persistent && slic_sweep_prof_buffer()) {
            long long h[16];
            HIP_TRY(hipMemcpy(h, slic_sweep_prof_buffer(), sizeof(h), hipMemcpyDeviceToHost));
            HIP_TRY(hipMemset(slic_sweep_prof_buffer(), 0, sizeof(h)));
            if (h[8] > 0) {
                fprintf(stderr, "[slic sweeps] per item, us:");
                for (int j = 0; j < 12; ++j)
                    if (j != 8) fprintf(stderr, " p%d=%.2f", j, (double)h[j] / (double)h[8] / 100.0);
                fprintf(stderr, "  (%lld items with accumulation)\n", h[8]);
            }
        }
        if (!used_persistent || *im->slic_fail_host == 0) break;       // (connectivity ended with a synchronisation)
        static const bool verbose = getenv("IMSEGM_DEBUG_SWEEPS") != nullptr;
        if (verbose) fprintf(stderr, "[slic sweeps] %d x %d, K = %d: handed back, code %d\n", H, W, K, *im->slic_fail_host);
        used_persistent = false;
    }
    ctx->end(sp_all);
    im->n_labels = n_labels;
    im->have_labels = true;
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
    im->graph_ready = false;
    return 0;
}

// diagnostic: number of 2-D connectivity passes of this process that left the tile path for the general one
long imsegm_debug_conn_general_runs(void) { return conn_general_runs(); }

void imsegm_debug_reload_env(void) { reload_knobs(); }

long imsegm_debug_gc_grid_fallbacks(void) { return gc_grid_fallbacks(); }

int imsegm_debug_slic_sweep_runs(long *persistent_runs_out, long *fallback_runs_out)
{
    slic_sweep_counters(persistent_runs_out, fallback_runs_out);
    return 0;
}

// skimage.segmentation._slic._enforce_label_connectivity_cython(segments, min_size, max_size, start_label) -- the second
// native call inside skimage.segmentation.slic (superpixels.py:61-63, enforce_connectivity=True) -- on a label map
// given by the caller; the result becomes the session's label map.  2-D image sessions and volume sessions.
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

static int stats_run(imsegm_image2d *im, const void *src, int dtype, double maxabs, int planar, int prescale, double mul,
                     double div, double *mean_out, double *energy_out, double *var_out, long plane_stride = -1);

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

static int stats_run(imsegm_image2d *im, const void *src, int dtype, double maxabs, int planar, int prescale, double mul,
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

int imsegm_image2d_lm_prepare(imsegm_image2d *im, const double *taps, int radius, const double *channel_mix)
{
    if (!im || bind(im->ctx)) return -1;
    if (im->dtype < 0) {
        set_error("no image uploaded");
        return -1;
    }
    if (radius < 0 || !taps || (!channel_mix && !im->is_volume)) {
        set_error("lm_prepare: bad arguments");
        return -1;
    }
    hipStream_t st = im->ctx->stream;
    // colour image: three channel planes of H x W; gray volume: its D slices, filtered independently (descriptors.py:981-994)
    const size_t np = im->is_volume ? im->n : 3 * im->n;
    if (im->tex_planes.ensure(np * 8) || im->labA.ensure(np * 8) || im->labB.ensure(np * 8)) return -1;
    if (im->tex_small.ensure(((size_t)radius + 1 + 9) * 8 + 1024 * 8 + 4096 + ((size_t)2 * radius + 64) * 8)) return -1;
    double *d_taps = im->tex_small.as<double>();
    double *d_mix = d_taps + radius + 1;
    double *d_full = d_taps + radius + 1 + 9 + 1024 + 512;        // behind the partial sums of the batteries
    HIP_TRY(hipMemcpyAsync(d_taps, taps, ((size_t)radius + 1) * 8, hipMemcpyHostToDevice, st));
    if (channel_mix) HIP_TRY(hipMemcpyAsync(d_mix, channel_mix, 9 * 8, hipMemcpyHostToDevice, st));
    HIP_TRY(hipStreamSynchronize(st));
    if (im->is_volume) {
        if (launch_texture_prepare_volume(im->img.p, im->dtype, im->D, im->H, im->W, d_taps, radius, im->tex_planes.as<double>(),
                                          im->labA.as<double>(), im->labB.as<double>(), st, d_full))
            return -1;
    } else if (launch_texture_prepare(im->img.p, im->dtype, im->H, im->W, d_taps, radius, d_mix, im->tex_planes.as<double>(),
                                      im->labA.as<double>(), im->labB.as<double>(), st, d_full)) {
        return -1;
    }
    im->tex_ready = true;
    return 0;
}

int imsegm_image2d_lm_battery(imsegm_image2d *im, const double *weights, int n_kernels, int radius, double clip,
                              double *sum_squares_out)
{
    if (!im || bind(im->ctx)) return -1;
    if (!im->tex_ready) {
        set_error("lm_battery: call imsegm_image2d_lm_prepare first");
        return -1;
    }
    hipStream_t st = im->ctx->stream;
    const size_t n = im->is_volume ? (im->n + 2) / 3 : im->n;          // the buffers below hold 3 * n values
    const int P = im->is_volume ? im->D : 3;
    const size_t S = 2 * (size_t)radius + 1;
    const size_t wbytes = S * S * n_kernels * 8;
    const size_t wpad = S * (S + 16) * n_kernels;                    // the row-padded copy the battery kernel reads (texture.hip)
    if (im->tex_resp.ensure(3 * n * 8 + wbytes + wpad * 8 + 1024 * 8 + 64)) return -1;
    double *resp = im->tex_resp.as<double>();
    double *d_w = resp + 3 * n;
    double *partial = d_w + S * S * n_kernels + wpad;
    double *d_sum = partial + 1024;
    HIP_TRY(hipMemcpyAsync(d_w, weights, wbytes, hipMemcpyHostToDevice, st));
    HIP_TRY(hipStreamSynchronize(st));
    int spx = im->ctx->begin(PG_TEX);
    if (launch_filter_battery(im->tex_planes.as<double>(), im->H, im->W, d_w, n_kernels, radius, clip, resp, partial, d_sum, st, P))
        return -1;
    im->ctx->end(spx);
    HIP_TRY(hipMemcpyAsync(sum_squares_out, d_sum, 8, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return 0;
}

static int take_placement(imsegm_image2d *im, int own_F, bool to_host, int *table_F, int *col0);

int imsegm_image2d_lm_features(imsegm_image2d *im, const double *weights, const int *n_kernels, int n_batteries, int radius, double clip,
                               int feature_mask, double *features_out)
{
    return imsegm_image2d_lm_features_sep(im, weights, n_kernels, nullptr, nullptr, nullptr, nullptr, n_batteries, radius, clip,
                                          feature_mask, features_out);
}

int imsegm_image2d_lm_features_sep(imsegm_image2d *im, const double *weights, const int *n_kernels, const int *dense_parity,
                                   const double *sep_taps, const int *sep_groups, const int *sep_rank, int n_batteries, int radius,
                                   double clip, int feature_mask, double *features_out)
{
    if (!im || bind(im->ctx)) return -1;
    if (wrong_kind(im, false)) return -1;
    if (!im->tex_ready || !im->have_labels) {
        set_error("lm_features: call imsegm_image2d_lm_prepare first, with a label map installed");
        return -1;
    }
    if (!n_kernels || n_batteries < 1 || feature_mask < 1 || feature_mask > 7 ||
        (sep_taps && (!sep_groups || !sep_rank))) {
        set_error("lm_features: bad arguments");
        return -1;
    }
    imsegm_ctx *ctx = im->ctx;
    hipStream_t st = ctx->stream;
    const size_t n = im->n;
    const int K = im->n_labels;
    const size_t S = 2 * (size_t)radius + 1;
    // per battery: the dense weights as the caller lays them out, room for the row-padded copy launch_filter_battery makes of them,
    // then the taps of its separable kernels (groups x rank components of 2 S doubles)
    std::vector<size_t> off((size_t)n_batteries + 1, 0), sep_off((size_t)n_batteries, 0);
    size_t dense_total = 0;
    for (int b = 0; b < n_batteries; ++b) {
        const int nk = n_kernels[b], ng = sep_taps ? sep_groups[b] : 0, rk = sep_taps ? sep_rank[b] : 0;
        if ((nk != 0 && nk != 1 && nk != 2 && nk != 4 && nk != 6 && nk != 8) || ng < 0 || ng > 2 || (ng > 0 && (rk < 1 || rk > 4)) ||
            (nk == 0 && ng == 0)) {
            set_error("filter battery: 0, 1, 2, 4, 6 or 8 dense kernels and up to 2 separable ones of rank 1..4 per battery");
            return -1;
        }
        sep_off[b] = off[b] + S * S * nk + S * (S + 16) * nk;
        off[b + 1] = sep_off[b] + (size_t)ng * rk * 2 * S;
        dense_total += S * S * nk;
    }
    if (dense_total > 0 && !weights) {
        set_error("lm_features: dense kernels without weights");
        return -1;
    }
    const size_t wtotal = off[n_batteries];
    // responses of up to ROUND consecutive batteries side by side: the separable kernels of a round share ONE launch (one load of
    // the input tile for all of them -- the five batteries of one sigma of the bank)
    const int ROUND = std::min(n_batteries, (int)SEP_MAX_JOBS);
    // (scratch of the sums of squares: 1024 partial sums of launch_response_sumsq, or one per workgroup and battery of the round when
    // the separable kernels -- the last writers of a response -- form them on the way)
    const size_t n_partial = std::max<size_t>(1024, sep_taps ? sep_sumsq_scratch(im->H, im->W, 3, radius, ROUND) : 0);
    if (im->tex_resp.ensure((3 * n * ROUND + wtotal + n_partial + (size_t)n_batteries + 8) * 8 + 64)) return -1;
    double *resp = im->tex_resp.as<double>();
    double *d_w = resp + 3 * n * ROUND;
    double *partial = d_w + wtotal;
    double *d_ssq = partial + n_partial;
    double *host = static_cast<double *>(ctx->stage(wtotal * 8));
    if (!host) {
        set_error("cannot allocate pinned staging memory");
        return -1;
    }
    {
        const double *src = weights, *ssrc = sep_taps;
        for (int b = 0; b < n_batteries; ++b) {
            const size_t cnt = S * S * n_kernels[b];
            if (cnt) memcpy(host + off[b], src, cnt * 8);
            src += cnt;
            const size_t scnt = sep_taps ? (size_t)sep_groups[b] * sep_rank[b] * 2 * S : 0;
            if (scnt) memcpy(host + sep_off[b], ssrc, scnt * 8);
            ssrc += scnt;
        }
    }
    HIP_TRY(hipMemcpyAsync(d_w, host, wtotal * 8, hipMemcpyHostToDevice, st));
    ctx->mark_stage_in_flight();
    // statistics scratch (as stats_run) and the K x F table
    const int nflags = ((feature_mask & 1) != 0) + ((feature_mask & 2) != 0) + ((feature_mask & 4) != 0);
    const int Fb = 3 * nflags, F = Fb * n_batteries;
    int table_F = F, col0 = 0;
    if (take_placement(im, F, features_out != nullptr, &table_F, &col0)) return -1;
    size_t fb = (size_t)K * (13 * 8 + 3 * 3 * 8 + 3 * 4) + 256;
    if (im->feat.ensure(fb) || im->featK.ensure((size_t)K * table_F * 8 + 64)) return -1;
    unsigned char *sb = im->feat.as<unsigned char>();
    long long *acc = reinterpret_cast<long long *>(sb); sb += (size_t)K * 13 * 8;
    double *d_mean = reinterpret_cast<double *>(sb); sb += (size_t)K * 3 * 8;
    double *d_energy = reinterpret_cast<double *>(sb); sb += (size_t)K * 3 * 8;
    double *d_var = reinterpret_cast<double *>(sb); sb += (size_t)K * 3 * 8;
    float *d_mean32 = reinterpret_cast<float *>(sb);
    for (int b0 = 0; b0 < n_batteries; b0 += ROUND) {
        const int cnt = std::min(ROUND, n_batteries - b0);
        int spx = ctx->begin(PG_TEX);
        SepJobs jobs;
        memset(&jobs, 0, sizeof(jobs));
        double *ssq_of_job[SEP_MAX_JOBS] = { nullptr };
        bool has_sep[SEP_MAX_JOBS] = { false };
        for (int j = 0; j < cnt; ++j) {
            const int b = b0 + j;
            double *rj = resp + (size_t)j * 3 * n;
            if (launch_battery_dense(im->tex_planes.as<double>(), im->H, im->W, d_w + off[b], n_kernels[b], radius, clip, rj, st, 3,
                                     dense_parity ? dense_parity[b] : 0))
                return -1;
            has_sep[j] = sep_taps && sep_groups[b] > 0;
            if (has_sep[j]) {
                ssq_of_job[jobs.n] = d_ssq + b;
                SepJob &q = jobs.job[jobs.n++];
                q.resp = rj; q.taps = d_w + sep_off[b]; q.groups = sep_groups[b]; q.rank = sep_rank[b]; q.merge = n_kernels[b] > 0 ? 1 : 0;
            }
        }
        const bool fused_ssq = jobs.n > 0 && sep_sumsq_scratch(im->H, im->W, 3, radius, jobs.n) > 0;
        if (launch_battery_sep(im->tex_planes.as<double>(), im->H, im->W, radius, clip, jobs, st, 3, fused_ssq ? partial : nullptr,
                               fused_ssq ? ssq_of_job : nullptr))
            return -1;
        for (int j = 0; j < cnt; ++j)
            if (!(fused_ssq && has_sep[j]) && launch_response_sumsq(resp + (size_t)j * 3 * n, 3 * n, partial, d_ssq + b0 + j, st)) return -1;
        ctx->end(spx);
        // |r| <= norm  =>  |r * mul / div| <= mul = log(1 + norm) / 0.03 < 2^15 for every finite norm: the bound the fixed-point
        // scales are chosen for, without the norm coming to the host (prescale 2: the kernels derive mul and div from *ssq)
        int sps = ctx->begin(PG_STATS);
        for (int j = 0; j < cnt; ++j) {
            const int b = b0 + j;
            if (launch_color_stats(resp + (size_t)j * 3 * n, IMSEGM_F64, im->labels.as<int32_t>(), im->H, im->W, K, 32768.0,
                                   (feature_mask & 2) != 0, acc, d_mean, d_energy, d_var, d_mean32, st, 1, 2, 1.0, 1.0, -1, d_ssq + b))
                return -1;
            if (launch_features_assemble(d_mean, d_energy, d_var, K, feature_mask, im->featK.as<double>(), st, table_F, col0 + b * Fb))
                return -1;
        }
        ctx->end(sps);
    }
    // called for the resident table (no host copy asked for): imsegm_image2d_segment reads it by feat_F; with a host copy the
    // table counts as consumed, as before
    im->feat_mask = features_out ? 0 : 8;
    im->feat_F = table_F;
    if (features_out) {
        HIP_TRY(hipMemcpyAsync(features_out, im->featK.p, (size_t)K * F * 8, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
    }
    return 0;
}

int imsegm_image2d_response_stats(imsegm_image2d *im, double mul, double div, double *mean_out, double *energy_out,
                                  double *var_out)
{
    if (!im || bind(im->ctx)) return -1;
    if (!im->tex_ready || !im->have_labels || im->tex_resp.cap < (im->is_volume ? im->n : 3 * im->n) * 8) {
        set_error("response_stats needs a filter response and a label map");
        return -1;
    }
    if (!(div != 0.0)) {
        set_error("response_stats: zero norm");
        return -1;
    }
    const double maxabs = fabs(mul);           // |r| <= norm = div  =>  |r * mul / div| <= |mul|
    if (!im->is_volume) return stats_run(im, im->tex_resp.p, IMSEGM_F64, maxabs, 1, 1, mul, div, mean_out, energy_out, var_out);
    // volume: the response is one plane of (D * H) x W read as all three channels (plane stride 0); K values each
    const int K = im->n_labels;
    std::vector<double> m((size_t)K * 3), e((size_t)K * 3), v((size_t)K * 3);
    const int keepH = im->H;
    im->H = im->D * keepH;
    int rc = stats_run(im, im->tex_resp.p, IMSEGM_F64, maxabs, 1, 1, mul, div, mean_out ? m.data() : nullptr,
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

int imsegm_image2d_get_response(imsegm_image2d *im, double *planes_out)
{
    if (!im || bind(im->ctx)) return -1;
    const size_t nv = im->is_volume ? im->n : 3 * im->n;
    if (im->tex_resp.cap < nv * 8) {
        set_error("no filter response");
        return -1;
    }
    HIP_TRY(hipMemcpyAsync(planes_out, im->tex_resp.p, nv * 8, hipMemcpyDeviceToHost, im->ctx->stream));
    HIP_TRY(hipStreamSynchronize(im->ctx->stream));
    return 0;
}

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

static ConnWork make_conn_work(imsegm_image2d *im)
{
    return conn_work_from(im->conn_i32.as<int32_t>(), im->conn_i32.cap, im->conn_u8.as<uint8_t>(), im->n);
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
    if (launch_label_cc(im->labels.as<int32_t>(), im->D, im->H, im->W, w.parent, w.newlabel, w.blocksum, w.counters, st)) return -1;
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

int imsegm_image2d_all_finite(imsegm_image2d *im, int *all_finite_out)
{
    if (!im || !all_finite_out || bind(im->ctx)) return -1;
    if (im->dtype < 0 || !im->img.p) {
        set_error("all_finite needs an uploaded image or volume");
        return -1;
    }
    *all_finite_out = 1;
    if (im->dtype == IMSEGM_U8) return 0;
    hipStream_t st = im->ctx->stream;
    if (ensure_small(im)) return -1;
    unsigned int *count = reinterpret_cast<unsigned int *>(im->small.as<unsigned char>() + 256);
    const size_t values = im->is_volume ? im->n : im->n * 3;
    if (launch_count_nonfinite(im->img.p, im->dtype, values, count, st)) return -1;
    unsigned int bad = 0;
    HIP_TRY(hipMemcpyAsync(&bad, count, sizeof(bad), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    *all_finite_out = bad == 0;
    return 0;
}

int imsegm_image2d_device_ptr(imsegm_image2d *im, int which, void **ptr_out)
{
    if (!im || !ptr_out) {
        set_error("null argument");
        return -1;
    }
    DevBuf *b = which == 0 ? &im->labels : which == 1 ? &im->gather_out_i : which == 2 ? &im->gather_out_f : nullptr;
    if (!b || !b->p) {
        set_error("device_ptr: buffer not available");
        return -1;
    }
    *ptr_out = b->p;
    return 0;
}

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
static int take_placement(imsegm_image2d *im, int own_F, bool to_host, int *table_F, int *col0)
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


// ---------------------------------------------------------------------------------------------------
// batched natives of features_cython.pyx (label histograms of windows, ray features of positions)
// ---------------------------------------------------------------------------------------------------
static int ctx_scratch(imsegm_ctx *ctx, size_t bytes, unsigned char **dev)
{
    if (ctx->gc_buf.ensure(bytes + 256)) return -1;
    *dev = ctx->gc_buf.as<unsigned char>();
    return 0;
}

int imsegm_assume_bg_on_boundary(imsegm_ctx *ctx, int32_t *segm_inout, int height, int width, const int32_t strips[16], int bg_label,
                                 int *boundary_label_out)
{
    if (bind(ctx)) return -1;
    if (!segm_inout || !strips || height <= 0 || width <= 0) {
        set_error("assume_bg_on_boundary: bad arguments");
        return -1;
    }
    for (int q = 0; q < 4; ++q)
        if (strips[4 * q] < 0 || strips[4 * q + 1] > height || strips[4 * q + 2] < 0 || strips[4 * q + 3] > width) {
            set_error("assume_bg_on_boundary: border strip outside the image");
            return -1;
        }
    hipStream_t st = ctx->stream;
    const size_t n = (size_t)height * width;
    unsigned char *dev = nullptr;
    if (ctx_scratch(ctx, n * 4 + 64, &dev)) return -1;
    int32_t *labels = reinterpret_cast<int32_t *>(dev);
    HIP_TRY(hipMemcpyAsync(labels, segm_inout, n * 4, hipMemcpyHostToDevice, st));
    // label range on the border (np.bincount sizes its result by the largest value and refuses negative ones)
    int32_t *mm_dev = nullptr;
    DevBuf &hb = ctx->aux_buf;
    if (hb.ensure(64)) return -1;
    mm_dev = hb.as<int32_t>();
    if (launch_boundary_minmax(labels, width, strips, mm_dev, st)) return -1;
    int32_t mm[2] = { 0, 0 };
    HIP_TRY(hipMemcpyAsync(mm, mm_dev, sizeof(mm), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    if (mm[0] > mm[1]) {
        set_error("assume_bg_on_boundary: empty border");
        return -1;
    }
    if (mm[0] < 0) {
        set_error("assume_bg_on_boundary: negative label on the border");
        return -1;
    }
    if (mm[1] >= (1 << 28)) {                 // (a histogram of 2^28 bins is 2 GB; INT32_MAX + 1 would overflow `nb`)
        set_error("assume_bg_on_boundary: border label too large for the border histogram");
        return -1;
    }
    const int nb = mm[1] + 1;
    if (hb.ensure((size_t)nb * 8 + 64)) return -1;
    unsigned long long *hist = hb.as<unsigned long long>();
    if (launch_boundary_hist(labels, width, strips, hist, nb, st)) return -1;
    std::vector<unsigned long long> h(nb);
    HIP_TRY(hipMemcpyAsync(h.data(), hist, (size_t)nb * 8, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    int best = 0;
    for (int i = 1; i < nb; ++i)
        if (h[i] > h[best]) best = i;                       // np.argmax: the first maximum
    if (boundary_label_out) *boundary_label_out = best;
    if (best != bg_label) {
        if (launch_swap_labels(labels, n, best, bg_label, st)) return -1;
        HIP_TRY(hipMemcpyAsync(segm_inout, labels, n * 4, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
    }
    return 0;
}

int imsegm_label_hist2d(imsegm_ctx *ctx, const int16_t *segm, int height, int width, const int32_t *windows, int n_windows,
                        const int16_t *struc_elem, int se_height, int se_width, int nb_labels, uint32_t *hist_out)
{
    if (bind(ctx)) return -1;
    if (!segm || !windows || !struc_elem || !hist_out || height < 1 || width < 1 || se_height < 1 || se_width < 1 || nb_labels < 1 ||
        n_windows < 0) {
        set_error("label_hist2d: bad arguments");
        return -1;
    }
    hipStream_t st = ctx->stream;
    auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
    const size_t b_seg = al((size_t)height * width * 2), b_win = al((size_t)n_windows * 24 + 8), b_se = al((size_t)se_height * se_width * 2);
    const size_t b_hist = al((size_t)n_windows * nb_labels * 4 + 8);
    unsigned char *dev;
    if (ctx_scratch(ctx, b_seg + b_win + b_se + b_hist, &dev)) return -1;
    HIP_TRY(hipMemcpyAsync(dev, segm, (size_t)height * width * 2, hipMemcpyHostToDevice, st));
    if (n_windows) HIP_TRY(hipMemcpyAsync(dev + b_seg, windows, (size_t)n_windows * 24, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(dev + b_seg + b_win, struc_elem, (size_t)se_height * se_width * 2, hipMemcpyHostToDevice, st));
    HIP_TRY(hipStreamSynchronize(st));                     // (pageable sources: the host buffers are free again)
    unsigned int *d_hist = reinterpret_cast<unsigned int *>(dev + b_seg + b_win + b_se);
    if (launch_label_hist2d(reinterpret_cast<int16_t *>(dev), height, width, reinterpret_cast<int32_t *>(dev + b_seg), n_windows,
                            reinterpret_cast<int16_t *>(dev + b_seg + b_win), se_height, se_width, nb_labels, d_hist, st))
        return -1;
    if (n_windows) HIP_TRY(hipMemcpyAsync(hist_out, d_hist, (size_t)n_windows * nb_labels * 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return 0;
}

int imsegm_ray_features_binary2d(imsegm_ctx *ctx, const int8_t *seg_binary, int height, int width, const int32_t *positions,
                                 int n_positions, const float *directions, int n_angles, int edge, float *ray_dist_out)
{
    if (bind(ctx)) return -1;
    if (!seg_binary || !positions || !directions || !ray_dist_out || height < 1 || width < 1 || n_positions < 0 || n_angles < 1 ||
        (edge != 1 && edge != -1)) {
        set_error("ray_features_binary2d: bad arguments (edge is 1 = up or -1 = down)");
        return -1;
    }
    hipStream_t st = ctx->stream;
    auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
    const size_t b_seg = al((size_t)height * width), b_pos = al((size_t)n_positions * 8 + 8), b_dir = al((size_t)n_angles * 8);
    const size_t b_out = al((size_t)n_positions * n_angles * 4 + 8);
    unsigned char *dev;
    if (ctx_scratch(ctx, b_seg + b_pos + b_dir + b_out, &dev)) return -1;
    HIP_TRY(hipMemcpyAsync(dev, seg_binary, (size_t)height * width, hipMemcpyHostToDevice, st));
    if (n_positions) HIP_TRY(hipMemcpyAsync(dev + b_seg, positions, (size_t)n_positions * 8, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(dev + b_seg + b_pos, directions, (size_t)n_angles * 8, hipMemcpyHostToDevice, st));
    HIP_TRY(hipStreamSynchronize(st));
    float *d_out = reinterpret_cast<float *>(dev + b_seg + b_pos + b_dir);
    if (launch_ray_features_binary2d(reinterpret_cast<int8_t *>(dev), height, width, reinterpret_cast<int32_t *>(dev + b_seg), n_positions,
                                     reinterpret_cast<float *>(dev + b_seg + b_pos), n_angles, edge, d_out, st))
        return -1;
    if (n_positions) HIP_TRY(hipMemcpyAsync(ray_dist_out, d_out, (size_t)n_positions * n_angles * 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return 0;
}


// ---------------------------------------------------------------------------------------------------
// 'median' and 'meanGrad' statistics on the resident image / volume and label map
// ---------------------------------------------------------------------------------------------------
int imsegm_image2d_median(imsegm_image2d *im, double *median_out)
{
    if (!im || bind(im->ctx)) return -1;
    if (!im->have_labels || im->dtype < 0 || !median_out) {
        set_error("median needs an uploaded image, a label map and an output");
        return -1;
    }
    hipStream_t st = im->ctx->stream;
    const int K = im->n_labels, C = im->is_volume ? 1 : 3;
    const size_t sb = median_scratch_bytes(im->n, K), ob = (size_t)K * C * 8;
    if (im->tex_resp.ensure(sb + ob + 256)) return -1;              // (the response buffer of the LM bank doubles as scratch)
    im->tex_ready = false;
    double *d_out = im->tex_resp.as<double>();
    unsigned char *scratch = im->tex_resp.as<unsigned char>() + ((ob + 255) & ~(size_t)255);
    if (launch_segment_median(im->img.p, im->dtype, C, im->n, im->labels.as<int32_t>(), K, scratch, sb, d_out, st)) return -1;
    HIP_TRY(hipMemcpyAsync(median_out, d_out, ob, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return 0;
}

int imsegm_image2d_mean_gradient(imsegm_image2d *im, double *mean_out)
{
    if (!im || bind(im->ctx)) return -1;
    if (!im->have_labels || im->dtype < 0 || !mean_out) {
        set_error("mean_gradient needs an uploaded image, a label map and an output");
        return -1;
    }
    if (im->H < 2 || im->W < 2) {
        set_error("Shape of array too small to calculate a numerical gradient, at least (edge_order + 1) elements are required.");
        return -1;
    }
    imsegm_ctx *ctx = im->ctx;
    hipStream_t st = ctx->stream;
    const int K = im->n_labels, C = im->is_volume ? 1 : 3;
    const size_t es = im->dtype == IMSEGM_U8 ? 1 : im->dtype == IMSEGM_F32 ? 4 : 8;
    if (im->tex_planes.ensure(im->n * C * es + 64)) return -1;      // gradient image, dtype of the source
    im->tex_ready = false;
    if (launch_gradient_image(im->img.p, im->tex_planes.p, im->dtype, im->D, im->H, im->W, C, st)) return -1;
    double maxabs = 255.0;
    if (im->dtype != IMSEGM_U8) {
        if (ensure_small(im)) return -1;
        unsigned long long *keys = im->small.as<unsigned long long>();
        double *minmax = reinterpret_cast<double *>(keys + 2);
        if (launch_minmax(im->tex_planes.p, im->dtype, im->n * C, keys, minmax, st)) return -1;
        double mm[2];
        HIP_TRY(hipMemcpyAsync(mm, minmax, 16, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        maxabs = std::max(fabs(mm[0]), fabs(mm[1]));
        if (!(maxabs < 1e300)) maxabs = 1e300;
    }
    std::vector<double> m((size_t)K * 3);
    int rc;
    if (im->is_volume) {
        const int keepH = im->H;
        im->H = im->D * keepH;
        rc = stats_run(im, im->tex_planes.p, im->dtype, maxabs, 1, 0, 1.0, 1.0, m.data(), nullptr, nullptr, 0);
        im->H = keepH;
        if (!rc)
            for (int k = 0; k < K; ++k) mean_out[k] = m[(size_t)k * 3];
    } else {
        rc = stats_run(im, im->tex_planes.p, im->dtype, maxabs, 0, 0, 1.0, 1.0, m.data(), nullptr, nullptr);
        if (!rc) memcpy(mean_out, m.data(), (size_t)K * 3 * 8);
    }
    return rc;
}

}  // extern "C"
