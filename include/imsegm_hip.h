/*
 * imsegm_hip.h -- C ABI of libimsegm_hip.so: the MI355X (gfx950) implementation of pyImSegm's
 * SLIC -> per-superpixel descriptors -> alpha-expansion GraphCut hot path.
 *
 * Conventions: every function returns 0 on success and a negative value on error; the message is
 * available from imsegm_last_error() (thread local).  All host arrays are C-contiguous and owned
 * by the caller; the library never keeps a host pointer past the return of a call.  The HIP
 * runtime is initialised lazily by the first call that needs it (never at load time), so the
 * library is safe to load before a fork() (reference: imsegm/utilities/experiments.py:392-403).
 *
 * Each entry point names the interface of the reference (Borda/pyImSegm @ /root/reference) or of
 * its third-party native dependency that it replaces.
 */
#ifndef IMSEGM_HIP_H
#define IMSEGM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define IMSEGM_API __attribute__((visibility("default")))

#define IMSEGM_U8 0
#define IMSEGM_F64 1
#define IMSEGM_F32 2

/* status of imsegm_image2d_segment / imsegm_image2d_graph_prepare when the FUSED form of the back half does not apply to this
 * label map on this device (its adjacency store does not fit the free memory, a label has more neighbours than a row of the
 * neighbour table): nothing is wrong with the session -- the caller builds the graph with imsegm_volume_graph /
 * imsegm_image2d_graph and cuts it with imsegm_cut_general_graph, the staged calls of imsegm/graph_cuts.py:660-747 */
#define IMSEGM_E_FUSED_PATH (-3)

typedef struct imsegm_ctx imsegm_ctx;           /* one device + one stream */
typedef struct imsegm_image2d imsegm_image2d;   /* device-resident state of one H x W image */

IMSEGM_API const char *imsegm_last_error(void);
IMSEGM_API int imsegm_version(void);
IMSEGM_API int imsegm_device_count(int *count_out);
/* PCI address of a device ("0000:c1:00.0", hipDeviceGetPCIBusId): with one process per GPU the host side of a rank belongs on the
 * NUMA node its GPU hangs off -- /sys/bus/pci/devices/<id>/numa_node (pyimsegm_amd.distributed.bind_to_device_numa_node).  The
 * reference's pool workers are not placed (imsegm/utilities/experiments.py:392-403); eight ranks each moving tens of GB/s over
 * the host link are. */
IMSEGM_API int imsegm_device_pci_bus_id(int device, char *id_out, int capacity);
/* free and total bytes of a device's memory right now (hipMemGetInfo): a gray-volume session keeps ~75 bytes per voxel resident
 * (imsegm_volume_*), so how many volumes of BASELINE configs[4] a device holds in flight is asked, not assumed */
IMSEGM_API int imsegm_device_mem_info(int device, size_t *free_bytes_out, size_t *total_bytes_out);
/* Optional, once per process and BEFORE the first call that touches a device: the number of hardware queues the HIP runtime maps
 * the streams of this process onto (the runtime's GPU_MAX_HW_QUEUES, default 4; with one stream per image in flight a fifth
 * stream shares a queue with another one and its kernels wait behind that image's -- bench.py asks for 8).  The runtime reads
 * the setting when it starts: returns 0 when the request was recorded, 1 when it can have no effect any more (this library has
 * already started the runtime) or the user has set GPU_MAX_HW_QUEUES himself (his value wins); nothing is changed then.
 * No reference counterpart (the reference's parallelism is a process pool, imsegm/utilities/experiments.py:392-403). */
IMSEGM_API int imsegm_init(int hardware_queues);
/* Debug / experiment switches (IMSEGM_* environment variables) are read once, at first use; this call reads them again
 * (tests flip them at run time).  No reference counterpart. */
IMSEGM_API void imsegm_debug_reload_env(void);

/* page-locked host memory: images / results living in it travel by DMA, asynchronously to the host (the Python layer
 * hands out numpy arrays backed by it: pyimsegm_amd._hip.pinned_empty) */
IMSEGM_API int imsegm_host_alloc(size_t bytes, void **ptr_out);
IMSEGM_API void imsegm_host_free(void *ptr);

/* plain device buffers and copies for the multi-GPU layer (pyimsegm_amd/distributed.py hands these pointers and the
 * context's stream to RCCL): hipMalloc / hipFree, hipSetDevice for the calling thread, hipMemcpyAsync (any direction)
 * on the context's stream */
IMSEGM_API int imsegm_device_alloc(int device, size_t bytes, void **ptr_out);
IMSEGM_API void imsegm_device_free(void *ptr);
IMSEGM_API int imsegm_set_device(int device);

IMSEGM_API int imsegm_ctx_create(int device, imsegm_ctx **ctx_out);
IMSEGM_API void imsegm_ctx_destroy(imsegm_ctx *ctx);
IMSEGM_API int imsegm_ctx_synchronize(imsegm_ctx *ctx);
IMSEGM_API int imsegm_ctx_stream(imsegm_ctx *ctx, void **hip_stream_out);
IMSEGM_API int imsegm_ctx_copy(imsegm_ctx *ctx, void *dst, const void *src, size_t bytes, int synchronize);

/* HIP-event timing of the kernels launched on the context's stream (bench.py roofline leg).
 * group: 0 = SLIC assignment(+accumulate) kernel, 1 = whole SLIC stage, 2 = connectivity,
 *        3 = colour statistics, 4 = adjacency + centres, 5 = alpha expansion, 6 = gathers,
 *        7 = SLIC pre-processing, 8 = class model + graph-cut terms, 9 = Leung-Malik filter batteries.  Returns accumulated milliseconds and launch
 *        count since reset. */
IMSEGM_API int imsegm_ctx_profile_enable(imsegm_ctx *ctx, int enable);
IMSEGM_API int imsegm_ctx_profile_reset(imsegm_ctx *ctx);
IMSEGM_API int imsegm_ctx_profile_get(imsegm_ctx *ctx, int group, double *total_ms_out, int *count_out);

/* ---------------------------------------------------------------------------------------------
 * device-resident pipeline object (keeps image, Lab planes, label map, ... in HBM between stages)
 * ------------------------------------------------------------------------------------------- */
IMSEGM_API int imsegm_image2d_create(imsegm_ctx *ctx, int height, int width, imsegm_image2d **img_out);
IMSEGM_API void imsegm_image2d_destroy(imsegm_image2d *img);

/* H2D copy of an H x W x 3 interleaved colour image (dtype IMSEGM_U8 / IMSEGM_F32 / IMSEGM_F64).  A page-locked source
 * (imsegm_host_alloc) is copied asynchronously: keep it unchanged until the next synchronising call (imsegm_image2d_slic). */
IMSEGM_API int imsegm_image2d_upload(imsegm_image2d *img, const void *host_pixels, int dtype);

/* Replaces skimage.segmentation.slic(image, n_segments, compactness, sigma, enforce_connectivity=True)
 * as called at imsegm/superpixels.py:61-63, with the min-max scaling of superpixels.py:53-54 folded
 * in (minmax_normalize: 1 = always, 0 = never, 2 = as the reference: unless min == 0 and max == 1).
 * taps_*: half kernels (taps[0] = centre, radius taps follow) of scipy.ndimage.gaussian_filter1d for
 * sigma / spacing per axis; radius < 0 disables the axis, radius > 16 (sigma / spacing > 4) is refused.  max_candidates: 0 = default (debug knob
 * that forces the kernel's global-memory fallback when small).  slic_zero: skimage's slic_zero=True
 * (SLICO, superpixels.py:63 `slic_zero=slico`).  Labels stay on the device. */
IMSEGM_API int imsegm_image2d_slic(imsegm_image2d *img, int minmax_normalize, int n_segments, double compactness,
                        const double *taps_z, int radius_z, const double *taps_y, int radius_y,
                        const double *taps_x, int radius_x, int max_iter, int enforce_connectivity,
                        double min_size_factor, double max_size_factor, int start_label,
                        int max_candidates, int slic_zero, int *n_labels_out);

/* copy the current label map to the host as int64 (dtype leaked by skimage, superpixels.py:69) */
IMSEGM_API int imsegm_image2d_get_labels(imsegm_image2d *img, int64_t *labels_out);
/* install an arbitrary label map (int32, values in [0, n_labels)) -- stage-level entry for the
 * descriptor / graph functions that take a user segmentation */
IMSEGM_API int imsegm_image2d_set_labels(imsegm_image2d *img, const int32_t *labels, int n_labels);
/* Diagnostic (no reference counterpart): how many 2-D connectivity passes of this process could not be finished by the
 * tile path of csrc/connectivity.hip and were redone by the general one (results are identical; tests use it to make sure
 * ordinary inputs stay on the fast path). */
IMSEGM_API long imsegm_debug_conn_general_runs(void);
/* Diagnostic (no reference counterpart): 2-D SLIC runs of this process whose sweeps 2..max_iter ran in the ONE persistent launch
 * of csrc/slic.hip (k_slic_sweeps), and how many of those gave the image back to the per-sweep launches (results are identical;
 * tests use it to make sure ordinary images stay on the persistent path, bench.py to know how many sweeps one launch covers). */
IMSEGM_API int imsegm_debug_slic_sweep_runs(long *persistent_runs_out, long *fallback_runs_out);
/* Diagnostic (no reference counterpart): graph cuts of this process that the grid-wide kernel (csrc/graphcut.hip
 * k_alpha_expansion_grid) gave up -- a workgroup of its grid was not resident within the bounded wait of its barrier, e.g. on a GPU
 * shared with another process -- and the single workgroup cut again from scratch (results are identical). */
IMSEGM_API long imsegm_debug_gc_grid_fallbacks(void);
/* Replaces skimage.segmentation._slic._enforce_label_connectivity_cython(segments, min_size, max_size, start_label)
 * (scikit-image 0.18; the connectivity pass of skimage.segmentation.slic, reached from imsegm/superpixels.py:61-63 and
 * :104-106 with enforce_connectivity=True) on a label map given by the caller: labels = host int32, one value per
 * pixel (voxel) of the session, raster order, all values >= start_label (skimage's mask label start_label - 1 does not
 * occur in the reference's calls and is not interpreted).  The result becomes the session's label map (imsegm_image2d_get_labels). */
IMSEGM_API int imsegm_image2d_enforce_connectivity(imsegm_image2d *img, const int32_t *labels, long min_size, long max_size,
                                                   int start_label, int *n_labels_out);
/* Replaces the per-pixel Python loop of imsegm/labeling.py:208-247 histogram_regions_labels_counts(slic, segm)
 * (called through histogram_regions_labels_norm at imsegm/pipelines.py:284 to label superpixels from an
 * annotation): hist_out[k * nb_annot + a] = number of pixels with resident label k and annotation a.
 * annot: host int32, one value per pixel (voxel) of the session; values outside [0, nb_annot) are not counted.
 * hist_out: host int64 [n_labels * nb_annot].  Works on image and volume sessions. */
IMSEGM_API int imsegm_image2d_label_hist(imsegm_image2d *img, const int32_t *annot, int nb_annot, int64_t *hist_out);
/* inspection for the parity tests: pre-processed Lab planes [3][H][W] and raw k-means assignment */
IMSEGM_API int imsegm_image2d_get_lab(imsegm_image2d *img, double *lab_out);
IMSEGM_API int imsegm_image2d_get_nearest(imsegm_image2d *img, int32_t *nearest_out);

/* Replaces imsegm.features_cython.computeColorImage2dMean / Energy / Variance + normColorFeatures
 * (imsegm/features_cython.pyx:59-141) on the uploaded image and the current label map.
 * Outputs are n_labels x 3 float64 (NULL = not wanted); variance uses the float32-rounded means as
 * descriptors.py:293 does. */
IMSEGM_API int imsegm_image2d_color_stats(imsegm_image2d *img, double *mean_out, double *energy_out, double *var_out);

/* Replaces make_graph_segm_connect_grid2d_conn4 (imsegm/superpixels.py:157-177) and
 * superpixel_centers (superpixels.py:205-242) on the current label map.
 * edges_out: capacity x 2 int32, pairs [a, b] with a < b ordered by (b, a); *n_edges_out receives the
 * real count (may exceed capacity: call again).  centres_out: n_labels x 2 (row, col), -1 for labels
 * without pixels; present_out: n_labels flags (the `vertices` of the reference). */
IMSEGM_API int imsegm_image2d_graph(imsegm_image2d *img, int32_t *edges_out, int edge_capacity, int *n_edges_out,
                         double *centres_out, uint8_t *present_out);

/* Replaces the LUT gathers graph_labels[slic] and proba[slic] (imsegm/pipelines.py:104,109).
 * Either output may be NULL. segm_out: H x W int32; soft_out: H x W x n_classes float64. */
IMSEGM_API int imsegm_image2d_gather(imsegm_image2d *img, const int32_t *graph_labels, const double *proba,
                          int n_classes, int32_t *segm_out, double *soft_out);

/* The whole Leung-Malik statistics of /root/reference/imsegm/descriptors.py:1041-1106 (compute_texture_desc_lm_img2d_clr: per
 * battery the filter responses, the maximum over the orientations, the clip, `response * (log(1 + norm) / 0.03) / norm` with
 * norm = the L2 norm over the three channels, then mean / std / energy per superpixel) in ONE call: the norm of a battery stays on
 * the device, the K x (3 * flags * n_batteries) table comes back once.  weights: the batteries one after the other, each
 * [kx][ky][kernel] with n_kernels[b] in {1, 2, 4, 8} kernels (flipped for a true convolution, last kernel repeated as padding:
 * what imsegm_image2d_lm_battery takes); feature_mask: 1 mean | 2 std | 4 energy, columns per battery in that order, three
 * channels each -- the column order of descriptors.py:1098-1103.  Needs imsegm_image2d_lm_prepare and a label map. */
IMSEGM_API int imsegm_image2d_lm_features(imsegm_image2d *img, const double *weights, const int *n_kernels, int n_batteries, int radius,
                                          double clip, int feature_mask, double *features_out);

/* The same with the separable kernels of the bank taken out of the dense sums: 28 of the 76 Leung-Malik kernels (the Gaussians, both
 * Laplacians of a Gaussian, the edge / bar filters at 0 and 90 degrees; descriptors.py:903-948) are matrices of rank 1 or 2, i.e.
 * sums of one or two products of a column and a row filter.  Battery b: n_kernels[b] dense kernels (0, 1, 2, 4, 6 or 8; laid out
 * as above) plus sep_groups[b] (0..2) separable kernels of sep_rank[b] (1..4) components each; sep_taps holds, battery after
 * battery, kernel after kernel, component after component, the 2 radius + 1 taps along x and then along y of the FLIPPED kernel
 * (the singular value folded into the y taps: K_flipped = sum_i y_i x_i^T).  The response of a battery is the maximum over all its
 * kernels, as before; the factorisation is the caller's (pyimsegm_amd._hip: numpy SVD, components above 1e-13 of the largest).
 * dense_parity[b] (or NULL): +1 / -1 when EVERY dense kernel of battery b is even / odd under the point reflection, K[-p] ==
 * +/- K[p] bit for bit (all bar / edge filters of the bank are) -- the sums are then formed over half the kernel, one addition
 * per pair of pixels serving all kernels of the battery; 0: no symmetry is assumed.  +2 / -2: the n_kernels[b] (2, 4, 6, 8) even /
 * odd kernels of side 33 are, in addition, mirror images of each other in pairs, Kb(dy, dx) = m Ka(dy, -dx) -- the orientations
 * theta and pi - theta of descriptors.py:924-928 -- and the weights block of the battery (still 33 * 33 * n_kernels[b] doubles)
 * holds, from its start, the table [x = 0..16][t = 0..16][WS of the pairs | WD of the pairs] with WS / WD = (Ka[t][16 + x] +/-
 * Ka[t][16 - x]) / 2 of the flipped kernel Ka, row t = 16 halved once more, followed by the signs m of the pairs: two additions and
 * two multiply-adds per four pixels and pair of kernels (csrc/texture.hip k_conv_battery_quad). */
IMSEGM_API int imsegm_image2d_lm_features_sep(imsegm_image2d *img, const double *weights, const int *n_kernels, const int *dense_parity,
                                              const double *sep_taps, const int *sep_groups, const int *sep_rank, int n_batteries,
                                              int radius, double clip, int feature_mask, double *features_out);
/* (features_out NULL, here only: the table stays on the device for imsegm_image2d_segment / imsegm_image2d_get_features and the call
 * returns without waiting for the kernels) */

/* Leung-Malik texture responses (imsegm/descriptors.py:951-1106, scipy.ndimage in the reference).
 * lm_prepare: planes = image - gaussian_filter(image, sigma) with `taps` = half kernel of
 *   scipy.ndimage.gaussian_filter1d(sigma) (taps[0] centre) for the two image axes and `channel_mix`
 *   the 3 x 3 matrix the same filter amounts to along the 3-element channel axis (descriptors.py:1078).
 * lm_battery: one filter battery = n_kernels (1, 2, 4, 8) square kernels of side 2*radius+1; `weights` is
 *   laid out [kx][ky][kernel] and already FLIPPED, i.e. response = max_k correlate(plane, weights[..k]) ==
 *   max_k ndimage.convolve(plane, kernel_k, mode='reflect') (descriptors.py:960-963), clipped at `clip`
 *   (:1088); returns the sum of squares over the three channels (for the norm of :1090).
 * response_stats: per-superpixel mean / energy / variance of (response * mul) / div (:1094-1096),
 *   float32 staging as imsegm_image2d_color_stats. */
IMSEGM_API int imsegm_image2d_lm_prepare(imsegm_image2d *img, const double *taps, int radius, const double *channel_mix);
IMSEGM_API int imsegm_image2d_lm_battery(imsegm_image2d *img, const double *weights, int n_kernels, int radius,
                                         double clip, double *sum_squares_out);
IMSEGM_API int imsegm_image2d_response_stats(imsegm_image2d *img, double mul, double div, double *mean_out,
                                             double *energy_out, double *var_out);
/* inspection for the parity tests: the current filter response as [3][H][W] planes */
IMSEGM_API int imsegm_image2d_get_response(imsegm_image2d *img, double *planes_out);

/* ---------------------------------------------------------------------------------------------
 * gray volumes D x H x W (the session type is shared; get_labels / set_labels / gather work on it)
 * ------------------------------------------------------------------------------------------- */
IMSEGM_API int imsegm_volume_create(imsegm_ctx *ctx, int depth, int height, int width, imsegm_image2d **vol_out);
/* The descriptors read the voxels as uploaded (float32 staging of features_cython.pyx); SLIC reads
 * (v + slic_offset) * slic_scale, the affine form of skimage.util.img_as_float for the source dtype
 * (uint8: 0, 1/255; floats: 0, 1). */
IMSEGM_API int imsegm_volume_upload(imsegm_image2d *vol, const void *host_voxels, int dtype, double slic_offset,
                                    double slic_scale);
/* Replaces skimage.segmentation.slic(im, n_segments, compactness, multichannel=False, spacing=space,
 * sigma=1) as called at imsegm/superpixels.py:104-106: gray supervoxels, anisotropic spacing[3] (z, y, x);
 * taps_* are the half kernels for sigma / spacing per axis. */
IMSEGM_API int imsegm_volume_slic(imsegm_image2d *vol, int n_segments, double compactness, const double *taps_z,
                                  int radius_z, const double *taps_y, int radius_y, const double *taps_x, int radius_x,
                                  const double *spacing, int max_iter, int enforce_connectivity, double min_size_factor,
                                  double max_size_factor, int start_label, int *n_labels_out);
/* Replaces skimage.measure.label(segments) (imsegm/superpixels.py:111): components of equal non-zero
 * value under full connectivity, numbered 1.. in raster order of their first voxel; 0 stays background.
 * (On the map imsegm_volume_slic has just written with enforce_connectivity -- every value > 0 one connected set -- this is a
 * renumbering by first voxels and is computed as one; any other map goes through the union-find.  Same result either way.) */
IMSEGM_API int imsegm_volume_label_cc(imsegm_image2d *vol, int *n_labels_out);
/* Replaces imsegm.features_cython.computeGrayImage3dMean / Energy / Variance (features_cython.pyx:144-219);
 * outputs n_labels float64 each (NULL = not wanted). */
IMSEGM_API int imsegm_volume_gray_stats(imsegm_image2d *vol, double *mean_out, double *energy_out, double *var_out);
/* Replaces make_graph_segm_connect_grid3d_conn6 and the 3-D branch of superpixel_centers
 * (imsegm/superpixels.py:180-242); centres_out is n_labels x 3 (z, y, x). */
IMSEGM_API int imsegm_volume_graph(imsegm_image2d *vol, int32_t *edges_out, int edge_capacity, int *n_edges_out,
                                   double *centres_out, uint8_t *present_out);

/* ---------------------------------------------------------------------------------------------
 * fused back half of the pipeline (no host round trip between the stages)
 * ------------------------------------------------------------------------------------------- */
/* Per-superpixel feature table of compute_image2d_color_statistic / compute_image3d_gray_statistic
 * (imsegm/descriptors.py:705-863) for the flags mean (1) | std (2) | energy (4): columns ordered mean, std, energy,
 * three columns each (colour channels; a gray volume repeats its single channel), NaN -> 0, -0 -> +0.  The table
 * stays resident for imsegm_image2d_segment; features_out (n_labels x 3*nflags float64) may be NULL. */
IMSEGM_API int imsegm_image2d_features_color(imsegm_image2d *img, int feature_mask, double *features_out);

/* The feature table of SEVERAL descriptor groups side by side, as compute_selected_features_color2d concatenates them
 * (imsegm/descriptors.py:1207-1270: the 'color' statistics, then the 'tLM' ones): the NEXT call of
 * imsegm_image2d_features_color / imsegm_image2d_lm_features_sep (with features_out NULL) writes its columns at `column` of a
 * resident table `total_columns` wide instead of a table of its own.  imsegm_image2d_segment evaluates the class model on that
 * table (F <= 256); the host reads it with imsegm_image2d_get_features (capacity_columns = its width, as a check). */
IMSEGM_API int imsegm_image2d_features_place(imsegm_image2d *img, int total_columns, int column);
IMSEGM_API int imsegm_image2d_get_features(imsegm_image2d *img, double *features_out, int capacity_columns);

/* class model evaluated on the device: sklearn Pipeline([StandardScaler,] GaussianMixture(covariance_type='full')) as
 * imsegm.graph_cuts.estim_class_model builds it (imsegm/graph_cuts.py:73-163).  The host passes what scikit-learn
 * itself precomputes per model (not per sample). */
typedef struct {
    int n_features, n_classes;
    const double *scaler_mean;    /* [F] StandardScaler.mean_ or NULL */
    const double *scaler_scale;   /* [F] StandardScaler.scale_ or NULL */
    const double *prec_chol;      /* [C][F][F] GaussianMixture.precisions_cholesky_ */
    const double *mu_proj;        /* [C][F]   means_[c] @ precisions_cholesky_[c] */
    const double *log_det;        /* [C]      _compute_log_det_cholesky */
    const double *log_weights;    /* [C]      log(weights_) */
    double const_term;            /* n_features * log(2 pi) */
} imsegm_gmm;

/* edge types of imsegm.graph_cuts.compute_edge_weights (imsegm/graph_cuts.py:574-657); `| IMSEGM_EDGE_SPATIAL_NORM`
 * divides the weights by the relative distance of the superpixel centres (:647-650: 'model', 'features', 'spatial') */
#define IMSEGM_EDGE_CONST 0
#define IMSEGM_EDGE_SPATIAL 1
#define IMSEGM_EDGE_MODEL_LT 2
#define IMSEGM_EDGE_MODEL_L1 3
#define IMSEGM_EDGE_MODEL_L2 4
#define IMSEGM_EDGE_FEATURES 5
#define IMSEGM_EDGE_SPATIAL_NORM 0x100

/* optional inspection outputs of imsegm_image2d_segment (parity tests, debug_visual): NULL pointers are skipped */
typedef struct {
    int edge_capacity;            /* in: rows of edges / edge_weights / edge_weights_int */
    int n_edges;                  /* out */
    int32_t *edges;               /* E x 2, a < b, ordered by (b, a) */
    double *edge_weights;         /* E, after clipping and edge_cost */
    int32_t *edge_weights_int;    /* E, the pyGCO integers */
    double *unary;                /* K x C */
    int32_t *unary_int;           /* K x C */
    double *centres;              /* K x ndim */
    int64_t *energy;              /* final integer energy of the expansion */
    int keep_soft_on_device;      /* in: compute proba[slic] into the session's buffer even when soft_out is NULL */
    /* in: narrow result formats (NOT the reference's dtypes -- explicit opt-in of the caller, SURVEY section 8f row 3: the
     * 100 MB float64 soft segmentation / the int32 class map dominate the device-to-host time once the kernels are fast) */
    int segm_u8;                  /* segm_out points to uint8 (height x width), classes must be < 256 */
    int soft_f32;                 /* soft_out points to float32 (height x width x n_classes) */
} imsegm_terms_debug;

/* Replaces, on the resident label map (and feature table): model.predict_proba (gmm != NULL; else `proba`, K x C, comes
 * from the host), compute_unary_cost, compute_edge_weights, compute_pairwise_cost's result `pairwise` (C x C, host),
 * gco.cut_general_graph(..., algorithm='expansion', n_iter=-1) (use_graphcut = 0: argmin of the unary cost,
 * graph_cuts.py:729-731), classes_[graph_labels] (classes_lut, may be NULL) and the gathers graph_labels[slic],
 * proba[slic] (imsegm/graph_cuts.py:660-747, imsegm/pipelines.py:96-109,232-240).  One stream, one synchronisation at
 * the end.  Outputs (each may be NULL): segm_out H x W int32, soft_out H x W x C float64, graph_labels_out K int32
 * (before classes_lut), proba_out K x C. */
IMSEGM_API int imsegm_image2d_segment(imsegm_image2d *img, const imsegm_gmm *gmm, const double *proba, int n_classes,
                                      const double *pairwise, int edge_type, double edge_cost, int use_graphcut,
                                      const int32_t *classes_lut, int32_t *segm_out, double *soft_out,
                                      int32_t *graph_labels_out, double *proba_out, imsegm_terms_debug *debug_out);

/* The part of imsegm_image2d_segment that depends on the label map alone -- make_graph_segm_connect_grid2d_conn4 / 3d_conn6 and
 * superpixel_centers (imsegm/superpixels.py:115-242) as edges, centres and the arcs the cut walks -- enqueued AHEAD of it and
 * without a synchronisation: the next imsegm_image2d_segment on the same label map finds the graph ready.  The gray-volume
 * pipeline fits its class model on the host between the descriptors and the cut (imsegm/pipelines.py:412-421); the graph is
 * built under that fit.  Any call that changes the label map drops the prepared graph.  May return IMSEGM_E_FUSED_PATH. */
IMSEGM_API int imsegm_image2d_graph_prepare(imsegm_image2d *img);

/* segment_color2d_slic_features_model_graphcut (imsegm/pipelines.py:160-241) for a colour image, colour statistics and a
 * mixture model the device evaluates, in ONE call: imsegm_image2d_upload + imsegm_image2d_slic (isotropic `taps`,
 * connectivity enforced with skimage's 0.5 / 3 size factors) + imsegm_image2d_features_color + imsegm_image2d_segment. */
IMSEGM_API int imsegm_image2d_run_color(imsegm_image2d *img, const void *host_pixels, int dtype, int minmax_normalize,
                                        int n_segments, double compactness, const double *taps, int radius, int max_iter,
                                        int start_label, int slic_zero, int feature_mask, const imsegm_gmm *gmm,
                                        int n_classes, const double *pairwise, int edge_type, double edge_cost,
                                        int use_graphcut, const int32_t *classes_lut, int32_t *segm_out, double *soft_out,
                                        int *n_labels_out);

/* ---------------------------------------------------------------------------------------------
 * several images of ONE size per launch chain
 * ------------------------------------------------------------------------------------------- */
/* Replaces the pool map of the reference's experiment driver over the images of a batch
 * (/root/reference/experiments_segmentation/run_segm_slic_model_graphcut.py:451-473 and :505-514 `segment_image_model` mapped through
 * imsegm/utilities/experiments.py:392-403 WrapExecuteSequence): up to `max_images` images of height x width go through
 * segment_color2d_slic_features_model_graphcut (imsegm/pipelines.py:160-241) TOGETHER -- every kernel of the chain is launched once
 * for the batch (image = blockIdx.z), with two host synchronisations per batch instead of two per image.  Results are those of
 * imsegm_image2d_run_color image by image, bit for bit. */
typedef struct imsegm_batch2d imsegm_batch2d;
IMSEGM_API int imsegm_batch2d_create(imsegm_ctx *ctx, int max_images, int height, int width, imsegm_batch2d **batch_out);
IMSEGM_API void imsegm_batch2d_destroy(imsegm_batch2d *batch);
/* host_pixels[i]: H x W x 3 interleaved image i (all of `dtype`); segm_out[i]: H x W int32 (a null entry, or segm_out == NULL:
 * that map stays on the device, imsegm_batch2d_device_ptr); n_labels_out: n_images superpixel counts or NULL.  Arguments as
 * imsegm_image2d_run_color; SLICO, a blur radius above 8 and host-evaluated class models take the single-image calls (an
 * error here, not a fallback). */
IMSEGM_API int imsegm_batch2d_run_color(imsegm_batch2d *batch, int n_images, const void *const *host_pixels, int dtype,
                                        int minmax_normalize, int n_segments, double compactness, const double *taps, int radius,
                                        int max_iter, int start_label, int feature_mask, const imsegm_gmm *gmm, int n_classes,
                                        const double *pairwise, int edge_type, double edge_cost, int use_graphcut,
                                        const int32_t *classes_lut, int32_t *const *segm_out, int *n_labels_out);
/* device address of a result of image `image` of the last batch: which = 0 label map, 1 segmentation (int32 H x W each) */
IMSEGM_API int imsegm_batch2d_device_ptr(imsegm_batch2d *batch, int image, int which, void **ptr_out);

/* The 'median' and 'meanGrad' statistics of compute_image2d_color_statistic / compute_image3d_gray_statistic
 * (imsegm/descriptors.py:420-455, 671-702, 766-770, 841-845) on the resident image (K x 3) or volume (K):
 * median of the pixel values per label and channel (NaN for labels without pixels, as np.median of an empty list);
 * mean over the label of np.sum(np.gradient(slice), axis=0) stored in the image's dtype (float32 staging as the other
 * means).  Both invalidate a prepared Leung-Malik state (they borrow its buffers). */
IMSEGM_API int imsegm_image2d_median(imsegm_image2d *img, double *median_out);
IMSEGM_API int imsegm_image2d_mean_gradient(imsegm_image2d *img, double *mean_out);

/* 1 when the uploaded image / volume holds no NaN and no inf (uint8: always).  The pipelines replace non-finite descriptor
 * values by zero (/root/reference/imsegm/pipelines.py:410 `features[np.isnan(features)] = 0`, descriptors.py:818), which the
 * resident statistics cannot reproduce from non-finite pixels: the host mirror asks before it takes the resident path (a
 * float64 sum over 10^9 voxels on the host costs more than the whole supervoxel stage). */
IMSEGM_API int imsegm_image2d_all_finite(imsegm_image2d *img, int *all_finite_out);

/* Device address of a result buffer of the session (valid until the next call that rewrites it):
 * which = 0: label map int32 H x W; 1: gathered segmentation int32 H x W; 2: gathered soft
 * segmentation float64 H x W x C.  For zero-copy hand-over to a collective library (RCCL) running on
 * the same HIP runtime; call imsegm_ctx_synchronize first. */
IMSEGM_API int imsegm_image2d_device_ptr(imsegm_image2d *img, int which, void **ptr_out);

/* ---------------------------------------------------------------------------------------------
 * stand-alone stages
 * ------------------------------------------------------------------------------------------- */
/* Replaces imsegm.labeling.assume_bg_on_boundary(segm, bg_label, boundary_size) for 2-D label images
 * (/root/reference/imsegm/labeling.py:719-753; called by the driver right after the pipeline,
 * experiments_segmentation/run_segm_slic_model_graphcut.py:373,422) including the border statistics of
 * imsegm.utilities.data_io.get_image2d_boundary_color (utilities/data_io.py:1025-1027: np.bincount over the four border
 * strips `image[:size, :], image[:, :size].T, image[-size:, :], image[:, -size:].T`, argmax = lowest label on ties).
 * segm_inout: host int32, height x width, non-negative on the border; strips: the four slices as {row0, row1, col0, col1}
 * (the caller resolves numpy's slice semantics); the label that dominates the border and `bg_label` are exchanged in place.
 * Returns < 0 with "negative label" when np.bincount would raise. */
IMSEGM_API int imsegm_assume_bg_on_boundary(imsegm_ctx *ctx, int32_t *segm_inout, int height, int width, const int32_t strips[16],
                                            int bg_label, int *boundary_label_out);

/* Replaces imsegm.features_cython.computeLabelHistogram2d (imsegm/features_cython.pyx:222-241; called per position through
 * descriptors.py:1411-1495 compute_label_hist_segm / cython_label_hist_seg2d) for a BATCH of windows of one label image:
 * window p = segm[y0 : y0 + h, x0 : x0 + w] against struc_elem[sy0 : sy0 + h, sx0 : sx0 + w], windows[p] = {y0, x0, h, w, sy0,
 * sx0}; hist_out[p][l] = pixels with label l (0 <= l < nb_labels) where the structuring element equals 1. */
IMSEGM_API int imsegm_label_hist2d(imsegm_ctx *ctx, const int16_t *segm, int height, int width, const int32_t *windows,
                                   int n_windows, const int16_t *struc_elem, int se_height, int se_width, int nb_labels,
                                   uint32_t *hist_out);
/* Replaces imsegm.features_cython.computeRayFeaturesBinary2d (features_cython.pyx:244-282; descriptors.py:1630-1660
 * cython_ray_features_seg2d) for a BATCH of positions (row, col): directions[a] = (sin, cos) of angle a divided by the
 * larger of their magnitudes, float32, formed by the caller as the .pyx forms them; edge 1 = 'up', -1 = 'down';
 * ray_dist_out[p][a] = distance to the edge along ray a, -1 if none (0 for every ray when 'up' starts inside). */
IMSEGM_API int imsegm_ray_features_binary2d(imsegm_ctx *ctx, const int8_t *seg_binary, int height, int width,
                                            const int32_t *positions, int n_positions, const float *directions, int n_angles,
                                            int edge, float *ray_dist_out);
/* Replaces gco.cut_general_graph(edges, edge_weights, unary_cost, pairwise_cost, n_iter,
 * algorithm='expansion') (gco-wrapper >= 3.0.8) as called at imsegm/graph_cuts.py:735-744.
 * edges: E x 2 int32 with edges[:,0] < edges[:,1]; edge_weights: E; unary: K x C; pairwise: C x C
 * symmetric.  Float costs are converted to integer energies exactly as pyGCO does.
 * Limits (an error, not a fallback): C <= 64 labels; n_iter < 0 = until convergence. */
IMSEGM_API int imsegm_cut_general_graph(imsegm_ctx *ctx, const int32_t *edges, int n_edges, const double *edge_weights,
                             const double *unary_cost, int n_sites, int n_labels, const double *pairwise_cost,
                             int n_iter, int32_t *labels_out, int64_t *energy_out);

#ifdef __cplusplus
}
#endif
#endif /* IMSEGM_HIP_H */
