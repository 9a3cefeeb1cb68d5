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
    k.pre_3pass = flag("IMSEGM_PRE_3PASS");
    k.separate_finalize = flag("IMSEGM_SEPARATE_FINALIZE");
    k.fuse_finalize = flag("IMSEGM_FUSE_FINALIZE");
    k.conn_general = flag("IMSEGM_CONN_GENERAL");
    k.gc_no_topo_regs = flag("IMSEGM_GC_NO_TOPO_REGS");
    k.sep_wide_tile = flag("IMSEGM_SEP_WIDE_TILE");
    k.cc_merge_full = flag("IMSEGM_CC_MERGE_FULL");
    k.label_general = flag("IMSEGM_LABEL_GENERAL");
    k.terms_one_workgroup = flag("IMSEGM_TERMS_ONE_WORKGROUP");
    k.adjacency_table = flag("IMSEGM_ADJACENCY_TABLE");
    k.brick_cap = num("IMSEGM_BRICK_CAP", 0);
    k.gc_lds_level = num("IMSEGM_GC_LDS_LEVEL", 4);
    k.gc_threads = num("IMSEGM_GC_THREADS", 0);
    k.gc_grid_min_sites = num("IMSEGM_GC_GRID_MIN_SITES", 0);
    k.gc_grid_blocks = num("IMSEGM_GC_GRID_BLOCKS", 0);
    k.gc_one_workgroup = flag("IMSEGM_GC_ONE_WORKGROUP");
    k.gc_grid_test_absent = flag("IMSEGM_GC_GRID_TEST_ABSENT");
    k.fused_bitmap_mb = num("IMSEGM_FUSED_BITMAP_MB", 0);
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
                      &im->gseg, &im->narrow };
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


}  // extern "C"
