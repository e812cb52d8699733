// train_kernels.h — launchers of the training-iteration kernels (train_loss.hip, train_post.hip, train_optim.hip),
// called from the C ABI in train_api.hip (include/surfel_train.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace surfel {

int ssim_blocks(int H, int W);
// window = the reference's window_size (odd, 3..15; loss_utils.py:43 default 11); false = unsupported size, nothing launched
bool launch_ssim_fwd(int window, int planes, int H, int W, const float* img, const float* gt, float* dmaps, float* partials, hipStream_t s);
bool launch_ssim_bwd(int window, int planes, int H, int W, const float* img, const float* gt, const float* dmaps, float c_l1, float c_ssim,
                     const float* g_l1_dev, const float* g_ssim_dev, float* grad_img, hipStream_t s);
void launch_reduce_partials(const float* partials, int groups, int n, int stride, float scale, float* out, hipStream_t s);
void launch_loss_finalize(const float* pa, int na, float scale_a, const float* pb, int nb, float scale_b, float lambda_dssim,
                          float lambda_normal, float lambda_dist, float* out, float* total_out, hipStream_t s);

int post_blocks(int H, int W);
void launch_post_fwd(int H, int W, const float* allmap, const float* cam, float ratio, float* maps, float* partials, hipStream_t s);
void launch_post_bwd(int H, int W, const float* allmap, const float* cam, float ratio, const float* gmaps, float c_normal, float c_dist,
                     const float* gscale_dev, float* gall, hipStream_t s);

// horizontally fused training loss (train_fused.hip): [3,H,W] image vs target (window 11) + allmap regularisers, one launch per direction
void launch_train_loss_fwd(int H, int W, const float* img, const float* gt, float* dmaps, float* partials, const float* allmap, const float* cam,
                           float ratio, float* post_partials, hipStream_t s);
// out6 != NULL: one extra workgroup of the launch computes the iteration's loss scalars from the forward's partial sums (deferred finalize)
void launch_train_loss_bwd(int H, int W, const float* img, const float* gt, const float* dmaps, float c_l1, float c_ssim, const float* g_dev,
                           float* grad_img, const float* allmap, const float* cam, float ratio, float c_normal, float c_dist, float* gall,
                           const float* ssim_partials, const float* post_partials, float lambda_dssim, float lambda_normal, float lambda_dist,
                           float* out6, float* total_out, hipStream_t s);

void launch_activate(int P, const float* theta, float* act, hipStream_t s);
void launch_adam(int P, float* theta, const float* grad, float* m, float* v, float* act, const float* lr, float beta1, float beta2, float eps,
                 float bc1, float bc2_sqrt, float grad_scale, int D, int N, const float* campos_all, const float* gcol_all, int parts,
                 hipStream_t s);
// densify_stats + adam (SH block rebuilt from the colour gradients + geometry sections) as one launch; g2d == NULL: no statistics
void launch_train_update(int P, float* theta, const float* grad, float* m, float* v, float* act, const float* lr, float beta1, float beta2,
                         float eps, float bc1, float bc2_sqrt, float grad_scale, int D, int N, const float* campos_all, const float* gcol_all,
                         const float* g2d, const int* radii, float* accum, float* denom, float* maxr, hipStream_t s);
void launch_sh_grad_gather(int P, int D, int N, const float* means3D, const float* campos_all, const float* gcol_all, float* dL_dsh,
                           hipStream_t s);
void launch_densify_stats(int P, const float* g2d, const int* radii, float* accum, float* denom, float* maxr, hipStream_t s);

// error plumbing shared with surfel_api.hip (thread-local message behind surfel_last_error())
int api_fail(int code, const char* what, hipError_t e = hipSuccess);

}  // namespace surfel
