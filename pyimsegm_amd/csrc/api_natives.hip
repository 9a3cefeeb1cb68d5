// api_natives.hip -- C ABI of the stand-alone natives: assume_bg_on_boundary, label histograms of windows, ray features, median, mean gradient
// (one of the files api.hip was split into in round 6: the C ABI of include/imsegm_hip.h by stage; the helpers they share are
// declared in session.h)
#include "session.h"

extern "C" {

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
