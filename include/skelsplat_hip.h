/*
 * skelsplat_hip.h -- C ABI of libskelsplat_hip.so (MI355X / gfx950).
 *
 * Drop-in boundary for the hot path of laurabragagnolo/SkelSplat: everything the reference binds through
 * pybind11 in submodules/diff-gaussian-rasterization-{h36m,panoptic,op}/ext.cpp:14-18
 * (rasterize_gaussians, rasterize_gaussians_backward, mark_visible; signatures rasterize_points.h:18-71),
 * submodules/fused-ssim/ext.cpp (fusedssim, fusedssim_backward) and submodules/simple-knn/ext.cpp (distCUDA2).
 * "DGR/" = submodules/diff-gaussian-rasterization-h36m/.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer into caller-owned, contiguous fp32 / int32 storage unless marked HOST;
 *    NULL means "not provided" (the reference uses data<float>() of an empty CPU tensor for that,
 *    DGR/diff_gaussian_rasterization_h36m/__init__.py:184-194);
 *  - the library never allocates, frees or synchronises; all work is enqueued on `stream`
 *    (a hipStream_t passed as void*), so calls are capturable into hipGraphs;
 *  - V views that share the Gaussian parameters and the image size are processed by ONE launch sequence
 *    (the reference renders one view per call; its loop, train.py:130-222, keeps the parameters fixed for
 *    `accumulation_steps` consecutive views, which is what makes the batch legal);
 *  - matrices are the reference's transposed 4x4 (row-major memory == column-major matrix), 16 floats per view;
 *  - return value: 0 ok, <0 invalid argument, >0 hipError_t; text via sks_last_error() (thread-local).
 */
#ifndef SKELSPLAT_HIP_H
#define SKELSPLAT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SKS_MAX_VIEWS 64      /* views per call */
#define SKS_MAX_CHANNELS 32   /* feature channels (reference: NUM_CHANNELS 15 / 17 / 19, config.h:15) */
#define SKS_SMALL_P 256       /* P <= SKS_SMALL_P: fused per-band path, no global binning */

/* flags */
#define SKS_ANTIALIASING 1u   /* raster_settings.antialiasing */
#define SKS_CLAMP01      2u   /* fold gaussian_renderer's rendered_image.clamp(0,1) (gaussian_renderer/__init__.py:129)
                                 into the forward store and its pass-through mask into the backward */
#define SKS_FORCE_BINNED 4u   /* use the binned (large-P) path even when P <= SKS_SMALL_P (tests) */
#define SKS_DEBUG_SYNC   8u   /* raster_settings.debug: hipStreamSynchronize + error check after each stage
                                 (CHECK_CUDA, DGR/cuda_rasterizer/auxiliary.h:178-185) */

#define SKS_NO_NT_STORES 16u  /* tuning: plain instead of non-temporal stores for the dense forward planes.  Non-temporal (the
                                 default) streams past the 256 MB Infinity Cache at the rate HBM takes writes; plain stores may
                                 stay in it: faster where a call rewrites the same ~cache-sized output step after step and
                                 nothing runs beside it (H36M: 46.5 -> 38-41 us), 40 % slower on outputs many times the cache.
                                 No result bit depends on it (rasterizer.autotune_fill_passes times both) */

/* tuning: bits 8..15 of `flags` = 4 KB passes per fill block of the fused forward (0 = automatic) */
#define SKS_RAW_PARAMS   32u   /* opacities / scales / rotations are the LEAF parameters (_opacity logits, _scaling
                                 log-scales, raw _rotation); sigmoid / exp / normalize (scene/gaussian_model.py:39-47)
                                 run inside the kernels (sks_geometry, sks_forward, sks_backward*) */
#define SKS_RAW_GRADS    128u   /* sks_backward with SKS_RAW_PARAMS: dL_dopacity / dL_dscales / dL_drotations are the gradients of the
                                  LEAF parameters (the activation Jacobians applied here) instead of those of the activated ones */
#define SKS_BIN_CLEAN    64u    /* accepted and ignored since sks_version 9: it used to promise that `binning` was as the previous
                                  sks_forward left it (per-tile counters zero again) so that a clearing launch could be skipped.
                                  The binned path no longer accumulates into the buffer -- every counter is written before it is
                                  read -- so a buffer needs no clearing and may hold anything on entry */
/* bits 16..18: binned path (P > SKS_SMALL_P), the views of a call are processed as (value + 1) VIEW GROUPS (at most 7, at most V):
   the binning kernels run once for all views, the forward's fill + composite launch and the backward's tile launch once per group.
   No result bit depends on it.  sks_forward_backward uses it to run group g's backward on the second stream while group g + 1's
   forward streams on the first; sks_backward must be given the value its sks_forward had (like every other flag). */
#define SKS_BIN_GROUPS_SHIFT 16
#define SKS_BIN_GROUPS(n) ((unsigned)(((n) - 1) & 7) << SKS_BIN_GROUPS_SHIFT)
#define SKS_FILL_LINEAR (1u << 21)  /* tuning/tests: forward fill blocks always in linear (pass-major) mode */
#define SKS_FILL_ROWS (1u << 22)    /* tuning/tests: row-aligned fill blocks whenever W % 4 == 0 */
/* bits 26..29: tuning, composite blocks per (view, Gaussian) of the small-path forward (0 = default 4) */
#define SKS_BWD_LDS_LIST (1u << 20) /* tests: use the LDS-list backward even when P <= 64 (default: wave-resident) */
/* bits 23..25: tuning/tests, workgroups per (view, Gaussian) of the wave-resident backward = 16 >> (value - 1)
   (0 = automatic: 16 for a few views, fewer and longer ones when V x P is large; the results do not depend on it) */
#define SKS_BWD_WG_SHIFT 23

const char* sks_last_error(void);
int sks_version(void);

/* Scratch sizes in bytes for one call (replaces the resizeFunctional callbacks, DGR/rasterize_points.cu:27-33).
 * geom: per-view per-Gaussian records kept for backward ("geomBuffer");
 * binning: the binned path's state ("binningBuffer" + "imgBuffer" of the reference), only used when P > SKS_SMALL_P or with
 *          SKS_FORCE_BINNED; bin_capacity = max (Gaussian, tile) pairs per view it must hold.  Since sks_version 8 it holds, per
 *          view: 48 B of entry records + 64 B of backward rows per pair of capacity (128 B in version 6), per 16x16 tile 2 KB of
 *          {final T, last contributor} for the backward + counters, range, descriptors, channel mask (~160 B) and a few KB of
 *          per-plane cover rows, per Gaussian 8 B -- e.g.
 *          0.65 GB for 8 views at 2048x2048 with a capacity of 400 000 pairs.  Sizes and layout are private to a library version:
 *          always ask the library that will be called;
 * accum:  backward partial-sum slots (plain scratch: no initialisation needed, contents undefined afterwards). */
int sks_scratch_bytes(int V, int P, int C, int W, int H, size_t bin_capacity,
                      size_t* geom_bytes, size_t* binning_bytes, size_t* accum_bytes);

/* Replaces _C.rasterize_gaussians (DGR/rasterize_points.cu:35-124 -> rasterizer_impl.cu:198-341).
 * features: (P,C) -- the reference reads them from `sh` with M == 1.
 * Outputs: out_color (V,C,H,W), out_invdepth (V,1,H,W), radii (V,P); every element is written
 * (no pre-zeroing needed).  out_color / out_invdepth must be 16-byte aligned (refused otherwise); 128-byte aligned
 * buffers -- every torch allocation is -- take the fastest fill (whole cache lines per pass).  Optional debug outputs final_T (V,H,W) / n_contrib (V,H,W) reproduce the
 * reference's ImageState (rasterizer_impl.h:52-60) for parity tests; pass NULL on the fast path.
 * num_rendered_dev (V ints, may be NULL): number of (Gaussian,tile) pairs, written on the binned path. */
int sks_forward(int V, int P, int C, int W, int H,
                const float* viewmatrix, const float* projmatrix,
                const float* tanfovx /*HOST V*/, const float* tanfovy /*HOST V*/,
                const float* means3D, const float* features, const float* opacities,
                const float* scales, const float* rotations, const float* cov3D_precomp,
                float scale_modifier, unsigned flags,
                float* out_color, float* out_invdepth, int* radii,
                void* geom, void* binning, size_t bin_capacity, int* num_rendered_dev,
                float* final_T, uint32_t* n_contrib, void* stream);

/* Replaces _C.rasterize_gaussians_backward (DGR/rasterize_points.cu:126-223 -> rasterizer_impl.cu:345-450).
 * geom / binning are the buffers sks_forward filled for the same inputs.  bg: C floats or NULL (zeros).
 * dL_dout_invdepth may be NULL (treated as zeros; the reference always receives a materialised zero grad, Q4).
 * Per-view gradients (V,P,...): dL_dmeans3D 3, dL_dmeans2D 3 (z = 0), dL_dopacity 1, dL_dscales 3, dL_drotations 4,
 * dL_dcov3D 6; dL_dfeatures (V,P,C) only when non-NULL (true dL/dfeature, SURVEY quirk Q5 is NOT reproduced).
 * dL_dscales / dL_drotations may be NULL when cov3D_precomp is given. */
int sks_backward(int V, int P, int C, int W, int H,
                 const float* viewmatrix, const float* projmatrix,
                 const float* tanfovx /*HOST V*/, const float* tanfovy /*HOST V*/,
                 const float* bg,
                 const float* means3D, const float* features, const float* opacities,
                 const float* scales, const float* rotations, const float* cov3D_precomp,
                 float scale_modifier, unsigned flags,
                 const int* radii, const void* geom, const void* binning, size_t bin_capacity,
                 const float* dL_dout_color, const float* dL_dout_invdepth,
                 void* accum,
                 float* dL_dmeans3D, float* dL_dmeans2D, float* dL_dopacity,
                 float* dL_dscales, float* dL_drotations, float* dL_dcov3D, float* dL_dfeatures,
                 float* dL_dmeans3D_mean /* optional (P,3): the mean of dL_dmeans3D over the V views, summed in view order --
                 what the reference's loop forms right after the backward (xyz.grad = accumulated_grads.mean(dim=0),
                 train.py:215-217); for V * P <= 256 it comes out of the same launch as the geometry backward */,
                 void* stream);

/* sks_forward + sks_backward of the same inputs as ONE call, for a caller whose upstream gradient does not depend on the image
 * this call renders (dL_dout_color / dL_dout_invdepth complete in `stream` order when the call is made: a known gradient, or the
 * previous evaluation's).  On the small path (P <= SKS_SMALL_P) the backward needs the forward's geometry records, not its image,
 * so with a second stream it runs BESIDE the dense forward instead of behind it: k_geom_fwd on `stream`; the backward's launches on
 * `aux_stream`, ordered behind the geometry kernel by an event; the fill + composite launch on `stream`; `stream` then waits for
 * the backward -- outputs and gradients are the caller's in `stream` order, exactly as after the two separate calls, and bit for
 * bit the same numbers.  The latency-bound backward (a few thousand short workgroups) hides under the HBM-bound forward: the H36M
 * step 67 -> ~57 us.  aux_stream NULL (or == stream, or the binned path, whose backward starts from what the forward's
 * compositor left per pixel, or more than 400 (view, Gaussian) pairs without SKS_FB_NO_JOIN -- there the backward's wavefronts
 * would cost the forward more than they hide: all 31 Panoptic views): the two calls one after the other.  Arguments as for sks_forward (without the two debug outputs)
 * followed by sks_backward's.  On the binned path with a second stream and SKS_BIN_GROUPS(n > 1) in `flags`, view group g's
 * backward runs on aux_stream beside group g + 1's forward (bit-identical; measured SLOWER than the plain sequence on the stress
 * scene, so it is not the default).  The library keeps a handful of hipEvents per host thread for the hand-overs (created by the
 * first combined call of a thread, re-created when the thread moves to another device, never destroyed: process-lifetime objects).
 * After an error behind the hand-over `stream` is made to wait for what aux_stream already holds, whatever fb_flags says.
 * fb_flags: SKS_FB_NO_JOIN = `stream` does NOT wait for the backward at the end: the caller has more to enqueue behind the
 * gradients on aux_stream -- a view-sharded step's collective on the joint gradients, which then also hides under the forward --
 * and makes `stream` wait for aux_stream itself (an event recorded on aux_stream, hipStreamWaitEvent on stream). */
#define SKS_FB_NO_JOIN 1u
int sks_forward_backward(int V, int P, int C, int W, int H,
                         const float* viewmatrix, const float* projmatrix,
                         const float* tanfovx /*HOST V*/, const float* tanfovy /*HOST V*/,
                         const float* means3D, const float* features, const float* opacities,
                         const float* scales, const float* rotations, const float* cov3D_precomp,
                         float scale_modifier, unsigned flags,
                         float* out_color, float* out_invdepth, int* radii,
                         void* geom, void* binning, size_t bin_capacity, int* num_rendered_dev,
                         const float* bg, const float* dL_dout_color, const float* dL_dout_invdepth, void* accum,
                         float* dL_dmeans3D, float* dL_dmeans2D, float* dL_dopacity,
                         float* dL_dscales, float* dL_drotations, float* dL_dcov3D, float* dL_dfeatures,
                         float* dL_dmeans3D_mean, void* stream, void* aux_stream, unsigned fb_flags);

/* xyz.grad = accumulated_grads.mean(dim=0) (train.py:215-217) on its own: mean over the V views of (V,P,3) joint gradients,
 * summed in view order.  shard_world = N > 1: the rows are where all_gather_into_tensor leaves them when view v is local view
 * v / N of rank v % N and every rank contributes ceil(V / N) rows (see sks_loop_adam_step) -- the exchange step of a
 * view-sharded caller is then all_gather + this one launch, with no re-ordering pass in between. */
int sks_mean_views(int V, int P, const float* dL_dmeans3D, int shard_world, float* mean_out /* (P,3) */, void* stream);

/* Replaces _C.mark_visible (DGR/rasterize_points.cu:225-244; checkFrustum rasterizer_impl.cu:54-66).
 * present: P bytes (bool). */
int sks_mark_visible(int P, const float* means3D, const float* viewmatrix, const float* projmatrix,
                     uint8_t* present, void* stream);

/* Debug / parity: copies the binned path's per-view lists out of `binning` in the reference's layout:
 * point_list (V,bin_capacity) uint32, ranges (V,Tx*Ty,2) uint32 (BinningState / ImageState::ranges). */
int sks_export_lists(int V, int W, int H, const void* binning, size_t bin_capacity,
                     uint32_t* point_list, uint32_t* ranges, void* stream);

/* Fused masked-L2 heat-map loss of the loop (utils/loss_utils.py:86-100 `l2_loss_gaussian`, called at train.py:150
 * on the clamped render; its autograd backward at train.py:161).  Per view v over n_per_view = C*H*W elements:
 *   N_v = #{gt > 0 or render > 0},  S_v = sum over that mask of (render - gt)^2  (loss_v = S_v / N_v),
 *   dL_unscaled = 2 (render - gt) on the mask, 0 elsewhere  (may be NULL: loss only).
 * The true gradient is dL_unscaled / N_v; everything downstream is linear in it, so callers scale the resulting
 * parameter gradients by 1 / N_v instead of re-reading the image.  sums: V x {S_v, N_v} doubles (zeroed by the call). */
int sks_masked_l2(int V, size_t n_per_view, const float* render, const float* gt, float* dL_unscaled, double* sums,
                  void* stream);
/* The same as a criterion: additionally loss[v] = S_v / N_v and scale[v] = 1 / N_v (mean != 0; an empty mask gives NaN like the
 * reference's `error[mask].mean()`), or S_v and 1 (mean == 0): what `l2_loss_gaussian(..., reduction="mean" / "sum")` returns
 * and what its backward scales dL_unscaled by -- formed on the device by one more tiny launch instead of five tensor ops. */
int sks_masked_l2_loss(int V, size_t n_per_view, const float* render, const float* gt, float* dL_unscaled, double* sums,
                       float* loss /* V */, float* scale /* V */, int mean, void* stream);

/* Pseudo-GT heat-maps of a scene (utils/general_utils.py:175-304 generate_heatmaps + normalize_heatmaps).  The
 * reference filters one 255 impulse per joint plane with cupyx gaussian_filter (V*J full-resolution calls) and
 * min-max normalises each plane; the filtered impulse is separable, so a plane is
 *   out[y][x] = (row[y] * col[x] - cmin) / den
 * with row = 255 * (1-D impulse response along y), col = the one along x, cmin = min(row) * min(col) and
 * den = max(row) * max(col) - cmin + 1e-8 (all fp32, in this order).  One pass, one write of (V,J,H,W).
 * row (V,J,H), col (V,J,W), cmin (V,J), den (V,J), out (V,J,H,W).  gt_totals (optional, V x 2 fp64): per view the sum of
 * out^2 and the count of out > 0 over all J planes -- what sks_gt_tile_stats would compute by reading the planes back. */
int sks_heatmaps(int V, int J, int W, int H, const float* row, const float* col, const float* cmin, const float* den,
                 float* out, double* gt_totals, void* stream);
/* The factors themselves, one launch (general_utils.py:189-289): lambda1/lambda2 of each joint's Gaussian in each view
 * by the reference's own transcription of the EWA projection, the scipy-'reflect' truncated (4 sigma) 1-D responses of
 * the 255 impulse at the truncated 2D detection, and the min-max constants.  means3D (J,3), scales (J,3) activated,
 * rotations (J,4) raw quaternions, poses_2d (V,J,2) pixel (x, y), viewmatrix (V,16) as for sks_forward,
 * tanfovx/tanfovy HOST arrays of V. */
int sks_heatmap_factors(int V, int J, int W, int H, const float* means3D, const float* scales, const float* rotations,
                        float scale_modifier, const float* poses_2d, const float* viewmatrix, const float* tanfovx,
                        const float* tanfovy, float* row, float* col, float* cmin, float* den,
                        int frames /* 1; > 1: V = frames x Vf views, parameters stacked (frames,J,..), see sks_loop_fused_step */,
                        const int* view_wh /* HOST V x {W,H} or NULL: per-view sizes; row / col keep the strides H / W */,
                        void* stream);
/* {sum gt^2, count gt > 0} per view (gt_totals, V x 2 fp64) of heat-maps that are NEVER WRITTEN: computed from the
 * separable factors alone (strides H / W, per-view sizes view_wh or NULL).  Together with hm_factors of
 * sks_backward_fused_loss / sks_loop_fused_step this replaces sks_heatmaps + the (V,J,H,W) planes on the sparse path. */
int sks_heatmap_totals(int V, int J, int W, int H, const float* row, const float* col, const float* cmin, const float* den,
                       const int* view_wh, double* gt_totals, void* stream);

/* Replaces fusedssim (submodules/fused-ssim/ssim.cu:368-404, binding ext.cpp): img1, img2, ssim_map and the three
 * optional partial-derivative maps (train == true) are (B,CH,H,W) fp32; "same" zero padding. */
int sks_fused_ssim_fwd(int B, int CH, int H, int W, float C1, float C2, const float* img1, const float* img2,
                       float* ssim_map, float* dm_dmu1, float* dm_dsigma1_sq, float* dm_dsigma12, void* stream);

/* Replaces fusedssim_backward (ssim.cu:406-444): dL_dimg1 (B,CH,H,W); only img1 is differentiable
 * (fused_ssim/__init__.py:32). */
int sks_fused_ssim_bwd(int B, int CH, int H, int W, float C1, float C2, const float* img1, const float* img2,
                       const float* dL_dmap, const float* dm_dmu1, const float* dm_dsigma1_sq, const float* dm_dsigma12,
                       float* dL_dimg1, void* stream);
/* The same backward when dL_dmap is what `.mean()` of the map hands back (fused_ssim/__init__.py:41): one value,
 * *dL_value (device memory) * dL_scale, on the image shrunk by `crop` pixels per side (5 for padding == "valid",
 * fused_ssim/__init__.py:13-14,24-26; 0 for "same") and zero outside.  No gradient image is materialised or read. */
int sks_fused_ssim_bwd_uniform(int B, int CH, int H, int W, const float* img1, const float* img2, const float* dL_value,
                               float dL_scale, int crop, const float* dm_dmu1, const float* dm_dsigma1_sq,
                               const float* dm_dsigma12, float* dL_dimg1, void* stream);
/* fused_ssim()'s `map.mean()` without the map (fused_ssim/__init__.py:34-41): *mean_out (device, fp32) = mean of the
 * SSIM map over the image shrunk by `crop` pixels per side (accumulated in fp64).  The three partial-derivative maps
 * are written when given (train == true), the map itself never.  scratch: SKS_SSIM_SCRATCH_BYTES of device memory that
 * is ZERO on entry and is left zero on exit (workgroups add to different slots, a one-wave kernel reduces and clears
 * them), so one buffer zeroed once serves every call on a stream. */
#define SKS_SSIM_SUM_SLOTS 64
#define SKS_SSIM_SCRATCH_BYTES (SKS_SSIM_SUM_SLOTS * 8)
int sks_fused_ssim_mean(int B, int CH, int H, int W, float C1, float C2, const float* img1, const float* img2, int crop,
                        float* dm_dmu1, float* dm_dsigma1_sq, float* dm_dsigma12, double* scratch, float* mean_out,
                        void* stream);

/* Replaces distCUDA2 (submodules/simple-knn/spatial.cu:15-26 -> SimpleKNN::knn simple_knn.cu:186-222):
 * points (P,3) -> mean squared distance to the 3 nearest neighbours (P). */
int sks_knn3_meandist2(int P, const float* points, float* mean_dist2, void* stream);
/* The same result (identical floats) for large clouds: exact uniform-grid search, O(P) for well-spread points
 * instead of the all-pairs sweep.  scratch: sks_knn3_scratch_bytes(P) bytes of device memory. */
size_t sks_knn3_scratch_bytes(int P);
int sks_knn3_meandist2_grid(int P, const float* points, float* mean_dist2, void* scratch, size_t scratch_bytes, void* stream);

/* Sparse fused training step (no dense image, no dense gradient).  In the loop the render is only ever compared with the
 * constant pseudo-GT heat-maps (train.py:148-152), and it is exactly zero outside the tiles some Gaussian rect covers,
 * so render + clamp + masked-L2 + backward can be evaluated on the covered tiles alone.  Where the render is zero the
 * loss only sees the heat-maps (mask gt > 0, error gt^2): a per-frame constant; where it is positive the exact term
 * replaces that constant.
 *  sks_gt_tile_stats (once per frame): per-view totals {sum of gt^2, count of gt > 0} (V x 2 doubles); optionally
 *      (tile_S / tile_N non-NULL) the same per (view, tile, channel) as (V, Ty*Tx, C) floats;
 *  sks_geometry: the geometry stage of sks_forward alone (fills `geom` and `radii`);
 *  sks_backward_fused_loss: like sks_backward, but takes the heat-maps `gt` (V,C,H,W) instead of dL/d(render):
 *      re-composites each covered pixel, forms 2 (clamp(r) - gt) on the mask {gt > 0 or r > 0} on the fly, and
 *      returns per-view {S, N} (loss_v = S/N) in loss_sums = gt_totals + the corrections of the pixels with a
 *      positive render; gradients are UNSCALED (multiply by 1/N_v, e.g. with sks_loop_pack_grads).  P <= 64.
 *      tile_S / tile_N are not read (may be NULL).
 *  Views of different image sizes in ONE launch sequence (H36M mixes 1000- and 1002-wide sensors,
 *  scene/dataset_readers.py:68-80): nothing dense is written on this path, so sks_geometry, sks_backward_fused_loss
 *  and sks_loop_fused_step accept view_wh = V x {W_v, H_v} (W, H arguments: the largest) and, for the heat-maps,
 *  gt_offsets = V offsets in floats from `gt` to view v's (C,H_v,W_v) planes inside one flat buffer.  NULL = all W x H
 *  and gt is one (V,C,H,W) tensor.  A `geom` filled with view_wh serves the fused-loss calls, not sks_forward. */
int sks_gt_tile_stats(int V, int C, int W, int H, const float* gt, float* tile_S, float* tile_N, double* totals, void* stream);
int sks_geometry(int V, int P, int C, int W, int H, const float* viewmatrix, const float* projmatrix,
                 const float* tanfovx /*HOST V*/, const float* tanfovy /*HOST V*/, const float* means3D,
                 const float* opacities, const float* scales, const float* rotations, const float* cov3D_precomp,
                 float scale_modifier, unsigned flags, int* radii, void* geom,
                 const int* view_wh /*HOST V x {W,H} or NULL: see below*/, int frames /* 1, or see sks_loop_fused_step */,
                 void* stream);
int sks_backward_fused_loss(int V, int P, int C, int W, int H, const float* viewmatrix, const float* projmatrix,
                            const float* tanfovx /*HOST V*/, const float* tanfovy /*HOST V*/, const float* bg,
                            const float* means3D, const float* features, const float* opacities, const float* scales,
                            const float* rotations, const float* cov3D_precomp, float scale_modifier, unsigned flags,
                            const int* radii, const void* geom, const float* gt, const float* tile_S, const float* tile_N,
                            const double* gt_totals, void* accum, float* dL_dmeans3D, float* dL_dmeans2D,
                            float* dL_dopacity, float* dL_dscales, float* dL_drotations, float* dL_dcov3D,
                            double* loss_sums, float* packed_raw_grads /* optional (V,P,11), see sks_loop_pack_grads:
                            with SKS_RAW_PARAMS the activation Jacobians and the 1/N_v scale are applied here */,
                            const int* view_wh /*HOST V x {W,H} or NULL*/, const size_t* gt_offsets /*HOST V or NULL*/,
                            const float* const* hm_factors /*HOST 4 device pointers or NULL, see below*/, void* stream);
/* hm_factors = {row (V,C,H), col (V,C,W), cmin (V,C), den (V,C)} (sks_heatmap_factors' outputs, strides H / W = this
 * call's W, H): the pseudo-GT is evaluated where it is needed, gt(v,c,y,x) = (row[y] * col[x] - cmin) / den -- the value
 * sks_heatmaps would have stored, bit for bit -- and `gt` / `gt_offsets` are not read (may be NULL); gt_totals then comes
 * from sks_heatmap_totals.  No heat-map plane exists anywhere on that path. */

/* Device-side tail of the multi-view loop (train.py:160-222), so that one accumulation group is a fixed launch
 * sequence with no host state (capturable into a hipGraph):
 * sks_loop_pack_grads: sks_backward's per-view gradients wrt the ACTIVATED tensors (V,P,..) -> packed (V,P,11) gradients
 *   wrt the RAW leaf parameters [xyz 3 | _scaling 3 | _rotation 4 | _opacity 1] through the exp / normalize / sigmoid
 *   Jacobians (scene/gaussian_model.py:39-47), times 1/N_v when loss_sums (V x {S,N} doubles of sks_masked_l2) is given.
 * sks_loop_adam_step: one optimiser step of the reference loop: slots[v] = grads[v].xyz + limb-symmetry gradient for the
 *   views in group_mask (utils/loss_utils.py:226-250, train.py:150-152,175), xyz.grad = mean over the V slots
 *   (train.py:215-217), scaling/rotation/opacity gradients of `last_view` (quirk Q7), exponential LR schedule evaluated at
 *   the stepping iteration (utils/general_utils.py:38-71, quirk Q9), torch.optim.Adam update (eps from the caller; the
 *   reference uses 1e-15, gaussian_model.py:218).  counters (device, 2 ints): [0] iteration, advanced by acc_steps,
 *   [1] Adam step count.  exp_avg / exp_avg_sq: (P,11) each, zero-initialised by the caller.  P <= SKS_SMALL_P. */
int sks_loop_pack_grads(int V, int P, const float* dL_dmeans3D, const float* dL_dscales, const float* dL_drotations,
                        const float* dL_dopacity, const float* raw_scaling, const float* raw_rotation,
                        const float* raw_opacity, const double* loss_sums, float* packed, void* stream);
int sks_loop_adam_step(int V, int P, const float* grads, float* slots, unsigned long long group_mask, int last_view,
                       float* xyz, float* scaling, float* rotation, float* opacity, float* exp_avg, float* exp_avg_sq,
                       int* counters, int acc_steps, const double* lr_sched /*HOST 5: init, final, delay_mult, delay_steps, max_steps*/,
                       const double* lrs /*HOST 3: scaling, rotation, opacity*/, const double* adam /*HOST 3: beta1, beta2, eps*/,
                       float lambda_consistency, const int* limb /*HOST 8 ints or NULL*/,
                       int shard_world /* layout of grads: 1 = (V,P,11) view-major; N > 1 = the buffer all_gather leaves when
                       view v is local view v / N of rank v % N and every rank contributes ceil(V / N) rows: (N, ceil(V/N), P, 11),
                       read in place -- the exchange step of the view-sharded loop needs no re-ordering pass */,
                       void* stream);

/* sks_loop_adam_step with the reference's early-stopping criterion ON THE DEVICE (training.early_stopping = opt_early_stopping:
 * utils/general_utils.py:467-491 fed at train.py:155, the break at train.py:182-233).  Before the step, one thread appends the
 * losses of the group's iterations -- loss_v = S_v / max(N_v, 1) + lambda x limb loss, fp32, in iteration order -- to the history
 * and tests the last `es_window` against the `es_window` before them (|difference| < es_tolerance, fp32).  When the criterion fires
 * at the k-th iteration of the group, only the first k views refresh their slots, view k's scaling / rotation / opacity gradients
 * win, the optimiser steps at once, `es_state[1]` (and *es_host_flag, pinned host memory, when given) receives that iteration --
 * and every later launch of this entry point on the same state does nothing: a caller keeps enqueueing groups without ever
 * reading a loss back and looks at the flag when it likes (no host synchronisation per group; the group is hipGraph-capturable).
 * es_state: 2 + 2 * es_window ints, zero at the start of a scene ([0] losses seen, [1] stopping iteration or 0, then the ring).
 * loss_sums: V x {S, N} doubles of the group's views (sks_masked_l2 / sks_backward_fused_loss).  shard_world = N > 1 and
 * loss_sums == NULL: every rank's block of `grads` is sks_loop_shard_floats(V, P, N) floats -- its ceil(V / N) x P x 11 gradient
 * rows (padded to an even count) followed by its views' {S, N} as doubles -- so gradients AND losses cross in the step's ONE
 * all_gather and every rank takes the identical decision. */
int sks_loop_adam_step_es(int V, int P, const float* grads, float* slots, unsigned long long group_mask, int last_view,
                          float* xyz, float* scaling, float* rotation, float* opacity, float* exp_avg, float* exp_avg_sq,
                          int* counters, int acc_steps, const double* lr_sched /*HOST 5*/, const double* lrs /*HOST 3*/,
                          const double* adam /*HOST 3*/, float lambda_consistency, const int* limb /*HOST 8 or NULL*/,
                          int shard_world, const double* loss_sums, int* es_state, int es_window, float es_tolerance,
                          int* es_host_flag /*pinned HOST int or NULL*/, void* stream);
size_t sks_loop_shard_floats(int V, int P, int shard_world);

/* torch.optim.Adam's update (no amsgrad, no weight decay; scene/gaussian_model.py:218 builds it with eps = 1e-15 over six parameter
 * groups, train.py:219 steps it) for up to 8 fp32 tensors in ONE launch:
 *   exp_avg.lerp_(grad, 1 - beta1); exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value = 1 - beta2);
 *   param.addcdiv_(exp_avg, exp_avg_sq.sqrt() / sqrt(1 - beta2^step) + eps, value = -lr / (1 - beta1^step)).
 * params / grads / exp_avg / exp_avg_sq: HOST arrays of n_tensors device pointers; numel, lr, step: HOST arrays (step = the
 * tensor's count AFTER this step, >= 1: torch keeps one per parameter).  What skelsplat_amd.optim.Adam.step() calls. */
int sks_adam_multi(int n_tensors, float* const* params, const float* const* grads, float* const* exp_avg, float* const* exp_avg_sq,
                   const long long* numel, const double* lr, const long long* step, double beta1, double beta2, double eps, void* stream);

/* One accumulation group of the sparse loop on ONE GPU in two launches (train.py:130-222 for acc_steps views):
 * the fused-loss compositing backward (as sks_backward_fused_loss) and a single-workgroup tail that runs the geometry
 * backward of every view, the optimiser step (as sks_loop_adam_step) and the geometry forward of the UPDATED parameters.
 * Precondition: `geom` / `radii` hold the geometry of the current parameters (sks_geometry with SKS_RAW_PARAMS, or the
 * previous sks_loop_fused_step); on return they hold the geometry of the updated ones.  xyz / scaling / rotation / opacity
 * are the RAW leaf parameters (updated in place); packed: (V,P,11) scratch (the group's raw-parameter gradients on
 * return); loss_sums: out, V x {S, N}.  P <= 64.  Results are bit-identical to the separate calls.
 *
 * frames > 1 batches INDEPENDENT frames (the reference optimises one frame after the other, train.py:74-99, and a
 * 17-Gaussian skeleton leaves most of the chip idle): V = frames x Vf views, view f*Vf + j is frame f's j-th view (the
 * camera rows are repeated per frame), and every frame owns a slice of the stacked tensors -- xyz/scaling (frames,P,3),
 * rotation (frames,P,4), opacity (frames,P), exp_avg/exp_avg_sq (frames,P,11), slots (frames,Vf,P,3), counters
 * (frames,2).  group_mask (bit j = view j of EVERY frame) and last_view are per frame (0 <= last_view < Vf); features,
 * the optimiser's hyper-parameters and the limb pairs are shared.  Each frame's results are bit-identical to running
 * it alone with frames = 1. */
int sks_loop_fused_step(int V, int P, int C, int W, int H, const float* viewmatrix, const float* projmatrix,
                        const float* tanfovx /*HOST V*/, const float* tanfovy /*HOST V*/, const float* features,
                        float scale_modifier, unsigned flags, int* radii, void* geom, const float* gt,
                        const double* gt_totals, void* accum, double* loss_sums, float* packed, float* slots,
                        unsigned long long group_mask, int last_view, float* xyz, float* scaling, float* rotation,
                        float* opacity, float* exp_avg, float* exp_avg_sq, int* counters, int acc_steps,
                        const double* lr_sched /*HOST 5*/, const double* lrs /*HOST 3*/, const double* adam /*HOST 3*/,
                        float lambda_consistency, const int* limb /*HOST 8 or NULL*/,
                        const int* view_wh /*HOST V x {W,H} or NULL*/, const size_t* gt_offsets /*HOST V or NULL*/,
                        int frames, const float* const* hm_factors /*HOST 4 or NULL, as for sks_backward_fused_loss*/,
                        void* stream);

/* Measurement hook used by bench.py (no reference counterpart; state per HOST THREAD, like the error text): while enabled, the dominant kernel of sks_forward
 * (kind 0: forward compositor) and of sks_backward (kind 1: backward compositor) is bracketed by hipEvents recorded
 * on the caller's stream (the small path's kernels carry the pair on their own dispatch, hipExtLaunchKernelGGL).  The low 16
 * bits of `on`: 1 = every launch, n > 1 = every n-th launch of each kind (a bracketed launch costs ~6 us of queue time, so
 * sampling keeps the measured loop undisturbed); bits 16-17: kinds to leave OUT (bit 16: kind 0, bit 17: kind 1).  sks_prof_read waits for the recorded events, returns
 * the summed kernel time in milliseconds and the number of BRACKETED launches since the last read, and resets them. */
int sks_prof_enable(int on);
/* Enqueues ONE wavefront that idles for `microseconds` (wall clock) on `stream`: a stand-in for the wire time of a latency-bound
 * collective, so that one GPU can measure what a rank of many sees when the exchange takes that long (bench.py rank_step_8gpu). */
int sks_prof_spin(double microseconds, void* stream);
/* bracketed launches of one kind collected since the last read (sks_prof_enable does not reset them: a caller may change the
 * sampling stride in the middle of a collection) */
int sks_prof_count(int kind, long long* launches);
int sks_prof_read(int kind, double* total_ms, long long* launches);
/* Same, plus the 10th / 50th / 90th percentile of the bracketed launch durations (milliseconds). */
int sks_prof_read_quantiles(int kind, double* q_ms /* 3 */, double* total_ms, long long* launches);

#ifdef __cplusplus
}
#endif
#endif
