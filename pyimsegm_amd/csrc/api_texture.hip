// api_texture.hip -- C ABI of the Leung-Malik texture stage (high-pass, batteries, response statistics)
// (one of the files api.hip was split into in round 6: the C ABI of include/imsegm_hip.h by stage; the helpers they share are
// declared in session.h)
#include "session.h"

extern "C" {

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


}  // extern "C"
