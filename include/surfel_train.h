/*
 * surfel_train.h — C ABI of the training-iteration kernels around the rasterizer,
 * part of libsurfel_hip.so (gfx950 only).  Same conventions as surfel_hip.h: plain DEVICE pointers and sizes,
 * `stream` = hipStream_t as void*, no allocation inside the library, return >= 0 or a negative SURFEL_E_* code
 * with the message in surfel_last_error().  Every output element is written by a kernel (no zero-filling needed).
 *
 * What each entry replaces in the reference (all host-side PyTorch there, ~60 small kernels per iteration):
 *   surfel_l1_ssim_*          utils/loss_utils.py:23-24 (l1_loss), :43-73 (ssim/_ssim, 11x11 sigma-1.5 window,
 *                             zero padding), combined as train.py:72-74
 *   surfel_render_post_*      gaussian_renderer/__init__.py:118-147 (allmap post-processing) +
 *                             utils/point_utils.py:9-37 (depths_to_points / depth_to_normal) and, in fused mode,
 *                             the regularisers of train.py:80-85
 *   surfel_reduce_partials    the `.mean()` reductions of the above, in a fixed (bit-reproducible) order
 *   surfel_loss_finalize      the scalar arithmetic of train.py:72-88 (loss, regularisers, total) in the same launch
 *   surfel_adam_step          scene/gaussian_model.py:95-115 (activations, backward folded in) +
 *                             torch.optim.Adam(eps=1e-15) over the six groups of :153-162, train.py:136-138
 *   surfel_activate           scene/gaussian_model.py:95-115 forward only (after (re)building the store)
 *   surfel_densify_stats      train.py:126-128 + scene/gaussian_model.py:405-407
 *   surfel_train_update       the two above in one launch
 */
#ifndef SURFEL_TRAIN_H
#define SURFEL_TRAIN_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Output tile edge of the SSIM kernels; partial-sum blocks per plane = ceil(W/32) * ceil(H/32). */
#define SURFEL_SSIM_TILE 32
/* Pixel tile edge of the post-processing kernels; partial-sum blocks = ceil(W/16) * ceil(H/16). */
#define SURFEL_POST_TILE 16

/*
 * L1 + SSIM forward over `planes` images of H x W (a [3,H,W] image = 3 planes; img = network output, gt = target).
 *   dmaps   [3, planes, H, W] or NULL: per-pixel dS/dmu1, dS/dE[x^2], dS/dE[xy] of the SSIM map S (saved for backward)
 *   partials [planes, nblk, 2]: per-workgroup (sum |img-gt|, sum S); reduce with surfel_reduce_partials.
 * Returns nblk.
 */
int surfel_l1_ssim_forward(int planes, int H, int W, const float* img, const float* gt, float* dmaps, float* partials,
                           void* stream);

/*
 * Backward: grad_img[planes,H,W] = c_l1 * g_l1 * sign(img-gt) + c_ssim * g_ssim * dSum(S)/dimg,
 * g_l1 = *g_l1_dev, g_ssim = *g_ssim_dev (device scalars, e.g. autograd's grad_output; NULL = 1).
 * For loss = (1-l)*mean|.| + l*(1-mean S):  c_l1 = (1-l)/N, c_ssim = -l/N, N = planes*H*W, both pointers = dL/dloss.
 */
int surfel_l1_ssim_backward(int planes, int H, int W, const float* img, const float* gt, const float* dmaps,
                            float c_l1, float c_ssim, const float* g_l1_dev, const float* g_ssim_dev, float* grad_img,
                            void* stream);

/* The same two with the reference's `window_size` argument (utils/loss_utils.py:43, default 11; sigma stays 1.5): odd sizes
 * 3..15, anything else returns SURFEL_E_INVALID.  The plain entry points above are these with window_size = 11. */
int surfel_l1_ssim_forward_w(int window_size, int planes, int H, int W, const float* img, const float* gt, float* dmaps,
                             float* partials, void* stream);
int surfel_l1_ssim_backward_w(int window_size, int planes, int H, int W, const float* img, const float* gt, const float* dmaps,
                              float c_l1, float c_ssim, const float* g_l1_dev, const float* g_ssim_dev, float* grad_img,
                              void* stream);

/*
 * Camera constants for the post-processing kernels, 24 floats on the device:
 *   [0..8]   A (row-major 3x3) = world_view_transform[:3,:3]: world normal = A * view normal
 *            (gaussian_renderer/__init__.py:123)
 *   [9..17]  K (row-major 3x3): ray direction of pixel (x,y) = (x,y,1) @ K, K = intrins^-1.T @ c2w[:3,:3].T
 *            (utils/point_utils.py:17-22)
 *   [18..20] ray origin c2w[:3,3]; [21..23] unused
 *
 * Forward: allmap[7,H,W] (rasterizer output) -> maps[9,H,W]:
 *   0 rend_alpha | 1-3 rend_normal (world) | 4 rend_dist | 5 surf_depth | 6-8 surf_normal (* alpha, detached)
 *   partials [nblk, 2] or NULL: per-workgroup (sum (1 - rend_normal . surf_normal), sum rend_dist)   (train.py:83-85)
 *   maps may be NULL when only the regulariser sums are wanted (the training loss: the backward recomputes from allmap); at
 *   least one of maps / partials must be given.
 * Returns nblk.
 */
int surfel_render_post_forward(int H, int W, const float* allmap, const float* cam, float depth_ratio, float* maps,
                               float* partials, void* stream);

/*
 * Backward: grad_allmap[7,H,W] from
 *   grad_maps[9,H,W] (upstream gradients of the 9 map channels) or NULL, PLUS the fused regulariser
 *   gscale * ( c_normal * sum(1 - rend_normal . surf_normal) + c_dist * sum(rend_dist) )   (c_* = lambda / (H*W)),
 *   gscale = *gscale_dev or 1 when NULL.  alpha inside surf_normal is detached (gaussian_renderer/__init__.py:147).
 * Pixels whose expected depth is 0/0 get zero gradient (the reference's autograd yields NaN there; the
 * rasterizer's backward never reads it because such pixels have no contributors).
 */
int surfel_render_post_backward(int H, int W, const float* allmap, const float* cam, float depth_ratio,
                                const float* grad_maps, float c_normal, float c_dist, const float* gscale_dev,
                                float* grad_allmap, void* stream);

/*
 * The training loss of one iteration (train.py:72-88) with both halves in ONE launch per direction: L1 + SSIM of the [3,H,W] image
 * against the target (window_size 11) and the allmap post-processing with the normal / distortion regulariser sums — the same
 * kernels as surfel_l1_ssim_* and surfel_render_post_* (maps = NULL, grad_maps = NULL), fused horizontally: the two halves share
 * no data, so their workgroups run side by side instead of as two dependent launches.  Same bits as the separate calls.
 *   forward : dmaps[3][3][H][W], ssim_partials[3 * ceil(W/32)*ceil(H/32)][2], post_partials[ceil(W/16)*ceil(H/16)][2]
 *             -> surfel_loss_finalize
 *   backward: g_dev = device scalar multiplying every constant (the upstream gradient of the total), may be NULL (= 1)
 */
int surfel_train_loss_forward(int H, int W, const float* img, const float* gt, float* dmaps, float* ssim_partials,
                              const float* allmap, const float* cam, float depth_ratio, float* post_partials, void* stream);
int surfel_train_loss_backward(int H, int W, const float* img, const float* gt, const float* dmaps, float c_l1, float c_ssim,
                               const float* allmap, const float* cam, float depth_ratio, float c_normal, float c_dist,
                               const float* g_dev, float* grad_img, float* grad_allmap,
                               /* deferred loss scalars: with out6 != NULL one extra workgroup of this launch does what
                                * surfel_loss_finalize does with the forward's two partial-sum arrays (same order, same bits), for a
                                * caller that reads the scalars only after the backward (a training loop) and saves that launch */
                               const float* ssim_partials, const float* post_partials, float lambda_dssim, float lambda_normal,
                               float lambda_dist, float* out6, float* total_out, void* stream);

/* out[g*stride + k] = scale * sum_i partials[(g*n + i)*stride + k], fixed summation order. groups*stride <= 65535. */
int surfel_reduce_partials(const float* partials, int groups, int n, int stride, float scale, float* out, void* stream);

/*
 * All loss scalars of one training iteration (train.py:72-88) from the two partial-sum arrays, one launch, fixed order:
 *   out6 = [Ll1, ssim, mean normal error, mean distortion, photometric, total]
 *   photometric = (1-lambda_dssim)*Ll1 + lambda_dssim*(1-ssim);  total = photometric + lambda_normal*[2] + lambda_dist*[3]
 * ssim_partials [n_ssim,2] over n_pixels_planes = planes*H*W elements; post_partials [n_post,2] over n_pixels = H*W, or NULL.
 * total_out (nullable): a second, separately owned copy of out6[5] (the differentiable output of an autograd node).
 */
int surfel_loss_finalize(const float* ssim_partials, int n_ssim, int n_pixels_planes, const float* post_partials, int n_post,
                         int n_pixels, float lambda_dssim, float lambda_normal, float lambda_dist, float* out6, float* total_out,
                         void* stream);

/*
 * The surfel parameter store: ONE flat fp32 buffer of 58 floats per surfel, planar by section
 *   xyz 3P | opacity P | scaling 2P | rotation 4P | sh 48P ([P,16,3]: coefficient 0 = f_dc, 1..15 = f_rest)
 * (raw, pre-activation values; the six Adam groups of scene/gaussian_model.py:153-160; the 10 geometry floats come
 * first so that view-parallel training all-reduces one contiguous 40 B/surfel prefix).  Gradients, Adam moments
 * and the all-reduce bucket use the same layout.  `act` [7P] = sigmoid(opacity) P | exp(scaling) 2P |
 * normalize(rotation) 4P — what the rasterizer consumes.
 */
int surfel_activate(int P, const float* theta, float* act, void* stream);

/*
 * One fused optimiser step.  grad[58P]: xyz/sh sections w.r.t. the raw parameters, opacity/scaling/rotation
 * sections w.r.t. the ACTIVATED values (what the rasterizer's backward produces) — the activation backward
 * (sigmoid', exp', normalize) is applied here.  grad is multiplied by grad_scale first (1/world after a SUM
 * all-reduce).  Adam exactly as torch.optim.Adam(betas, eps, no weight decay, no amsgrad) at step `t` (1-based),
 * per-group learning rates lr[6] = xyz, f_dc, f_rest, opacity, scaling, rotation (host array).
 * Writes theta, m, v in place and the new activated values to act.
 * SH block: with gcol_all == NULL its gradients are read from grad like the rest.  With gcol_all [N,P,3] (+ campos_all [N,3],
 * active degree D) they are rebuilt in registers as in surfel_sh_grad_gather — the 192 B/surfel SH gradient is then never
 * written (surfel_rasterize_backward accepts dL_dsh == NULL) nor read; N = 1 for single-GPU training.
 * parts: 3 = the whole step; 1 = SH block only, 2 = geometry sections (xyz, opacity, scaling, rotation + activations) only —
 * two calls with the same t make one step (1 before 2: the SH rebuild reads the positions of the forward), which lets a
 * view-parallel trainer update the SH block as soon as the colour all-gather has landed, while the geometry all-reduce is in flight.
 */
int surfel_adam_step(int P, float* theta, const float* grad, float* m, float* v, float* act, const float* lr,
                     float beta1, float beta2, float eps, int t, float grad_scale,
                     int D, int N, const float* campos_all, const float* gcol_all, int parts, void* stream);

/*
 * surfel_densify_stats + surfel_adam_step(parts = 3, SH block rebuilt from the colour gradients) as ONE launch, for the iterations in
 * which nothing is rebuilt between the two (train.py:126-138 without the densification branch): same results to the bit, two
 * dependent launch boundaries and the statistics' all-latency kernel fewer.  dL_dmeans2D == NULL: no statistics (iterations behind
 * densify_until_iter).  gcol_all / campos_all are required (N >= 1).
 */
int surfel_train_update(int P, float* theta, const float* grad, float* m, float* v, float* act, const float* lr,
                        float beta1, float beta2, float eps, int t, float grad_scale,
                        int D, int N, const float* campos_all, const float* gcol_all,
                        const float* dL_dmeans2D, const int* radii, float* grad_accum, float* denom, float* max_radii, void* stream);

/*
 * View-parallel training (new; the reference is single-GPU): the SH gradient of the summed loss rebuilt from every
 * rank's clamp-masked dL/dcolour instead of all-reducing 192 B/surfel:
 *   dL_dsh[P,16,3] = sum_{r<N} basis(normalize(means3D - campos_all[r])) (x) gcol_all[r, :, :]      (rank order, D = active degree)
 * gcol_all [N,P,3] = all-gather of surfel_rasterize_backward's dL_dcolors (in SH mode: masked by the forward's clamp);
 * campos_all [N,3] = the camera centres of the N views of this step.  Basis as utils/sh_utils.py:57-112.
 */
int surfel_sh_grad_gather(int P, int D, int N, const float* means3D, const float* campos_all, const float* gcol_all,
                          float* dL_dsh, void* stream);

/*
 * Densification statistics of one rendered view: for surfels with radii > 0:
 *   grad_accum[i] += || dL_dmeans2D[i, 0:3] ||, denom[i] += 1, max_radii[i] = max(max_radii[i], radii[i]).
 */
int surfel_densify_stats(int P, const float* dL_dmeans2D, const int* radii, float* grad_accum, float* denom,
                         float* max_radii, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SURFEL_TRAIN_H */
