// train_api.hip — C ABI of the training-iteration kernels (include/surfel_train.h).  Host code only: argument
// checks, launches on the caller's stream, hipGetLastError.  No allocation, no synchronisation.
#include <cmath>

#include <hip/hip_runtime.h>

#include "../../include/surfel_debug.h"
#include "../../include/surfel_train.h"
#include "train_kernels.h"

using namespace surfel;

namespace {
inline int launched(const char* what) {
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : api_fail(SURFEL_E_HIP, what, e);
}
}  // namespace

extern "C" {

int surfel_l1_ssim_forward_w(int window_size, int planes, int H, int W, const float* img, const float* gt, float* dmaps, float* partials,
                             void* stream) {
    if (planes <= 0 || H <= 0 || W <= 0 || !img || !gt || !partials) return api_fail(SURFEL_E_INVALID, "l1_ssim_forward: bad arguments");
    if (!launch_ssim_fwd(window_size, planes, H, W, img, gt, dmaps, partials, static_cast<hipStream_t>(stream)))
        return api_fail(SURFEL_E_INVALID, "l1_ssim_forward: window_size must be odd and in 3..15");
    const int rc = launched("ssim_fwd_kernel");
    return rc < 0 ? rc : ssim_blocks(H, W);
}

int surfel_l1_ssim_backward_w(int window_size, int planes, int H, int W, const float* img, const float* gt, const float* dmaps, float c_l1,
                              float c_ssim, const float* g_l1_dev, const float* g_ssim_dev, float* grad_img, void* stream) {
    if (planes <= 0 || H <= 0 || W <= 0 || !img || !gt || !dmaps || !grad_img) return api_fail(SURFEL_E_INVALID, "l1_ssim_backward: bad arguments");
    if (!launch_ssim_bwd(window_size, planes, H, W, img, gt, dmaps, c_l1, c_ssim, g_l1_dev, g_ssim_dev, grad_img, static_cast<hipStream_t>(stream)))
        return api_fail(SURFEL_E_INVALID, "l1_ssim_backward: window_size must be odd and in 3..15");
    return launched("ssim_bwd_kernel");
}

int surfel_l1_ssim_forward(int planes, int H, int W, const float* img, const float* gt, float* dmaps, float* partials, void* stream) {
    return surfel_l1_ssim_forward_w(11, planes, H, W, img, gt, dmaps, partials, stream);
}

int surfel_l1_ssim_backward(int planes, int H, int W, const float* img, const float* gt, const float* dmaps, float c_l1, float c_ssim,
                            const float* g_l1_dev, const float* g_ssim_dev, float* grad_img, void* stream) {
    return surfel_l1_ssim_backward_w(11, planes, H, W, img, gt, dmaps, c_l1, c_ssim, g_l1_dev, g_ssim_dev, grad_img, stream);
}

int surfel_render_post_forward(int H, int W, const float* allmap, const float* cam, float depth_ratio, float* maps, float* partials,
                               void* stream) {
    if (H <= 0 || W <= 0 || !allmap || !cam || (!maps && !partials)) return api_fail(SURFEL_E_INVALID, "render_post_forward: bad arguments");
    launch_post_fwd(H, W, allmap, cam, depth_ratio, maps, partials, static_cast<hipStream_t>(stream));
    const int rc = launched("post_fwd_kernel");
    return rc < 0 ? rc : post_blocks(H, W);
}

int surfel_render_post_backward(int H, int W, const float* allmap, const float* cam, float depth_ratio, const float* grad_maps,
                                float c_normal, float c_dist, const float* gscale_dev, float* grad_allmap, void* stream) {
    if (H <= 0 || W <= 0 || !allmap || !cam || !grad_allmap) return api_fail(SURFEL_E_INVALID, "render_post_backward: bad arguments");
    launch_post_bwd(H, W, allmap, cam, depth_ratio, grad_maps, c_normal, c_dist, gscale_dev, grad_allmap, static_cast<hipStream_t>(stream));
    return launched("post_bwd_kernel");
}

int surfel_train_loss_forward(int H, int W, const float* img, const float* gt, float* dmaps, float* ssim_partials, const float* allmap,
                              const float* cam, float depth_ratio, float* post_partials, void* stream) {
    if (H <= 0 || W <= 0 || !img || !gt || !dmaps || !ssim_partials || !allmap || !cam || !post_partials) return api_fail(SURFEL_E_INVALID, "bad arguments");
    launch_train_loss_fwd(H, W, img, gt, dmaps, ssim_partials, allmap, cam, depth_ratio, post_partials, static_cast<hipStream_t>(stream));
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : api_fail(SURFEL_E_HIP, "train_loss_fwd", e);
}

int surfel_train_loss_backward(int H, int W, const float* img, const float* gt, const float* dmaps, float c_l1, float c_ssim, const float* allmap,
                               const float* cam, float depth_ratio, float c_normal, float c_dist, const float* g_dev, float* grad_img,
                               float* grad_allmap, const float* ssim_partials, const float* post_partials, float lambda_dssim,
                               float lambda_normal, float lambda_dist, float* out6, float* total_out, void* stream) {
    if (H <= 0 || W <= 0 || !img || !gt || !dmaps || !allmap || !cam || !grad_img || !grad_allmap) return api_fail(SURFEL_E_INVALID, "bad arguments");
    if (out6 && (!ssim_partials || !post_partials)) return api_fail(SURFEL_E_INVALID, "train_loss_backward: deferred loss scalars need both partial-sum arrays");
    launch_train_loss_bwd(H, W, img, gt, dmaps, c_l1, c_ssim, g_dev, grad_img, allmap, cam, depth_ratio, c_normal, c_dist, grad_allmap,
                          ssim_partials, post_partials, lambda_dssim, lambda_normal, lambda_dist, out6, total_out, static_cast<hipStream_t>(stream));
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : api_fail(SURFEL_E_HIP, "train_loss_bwd", e);
}

int surfel_reduce_partials(const float* partials, int groups, int n, int stride, float scale, float* out, void* stream) {
    if (!partials || !out || groups <= 0 || n <= 0 || stride <= 0 || stride > 65535 || groups > 65535)
        return api_fail(SURFEL_E_INVALID, "reduce_partials: bad arguments");
    launch_reduce_partials(partials, groups, n, stride, scale, out, static_cast<hipStream_t>(stream));
    return launched("reduce_partials_kernel");
}

int surfel_loss_finalize(const float* ssim_partials, int n_ssim, int n_pixels_planes, const float* post_partials, int n_post, int n_pixels,
                         float lambda_dssim, float lambda_normal, float lambda_dist, float* out6, float* total_out, void* stream) {
    if (!ssim_partials || n_ssim <= 0 || n_pixels_planes <= 0 || !out6 || (post_partials && (n_post <= 0 || n_pixels <= 0)))
        return api_fail(SURFEL_E_INVALID, "loss_finalize: bad arguments");
    launch_loss_finalize(ssim_partials, n_ssim, 1.f / (float)n_pixels_planes, post_partials, n_post, post_partials ? 1.f / (float)n_pixels : 0.f,
                         lambda_dssim, lambda_normal, lambda_dist, out6, total_out, static_cast<hipStream_t>(stream));
    return launched("loss_finalize_kernel");
}

int surfel_activate(int P, const float* theta, float* act, void* stream) {
    if (P < 0 || (P > 0 && (!theta || !act))) return api_fail(SURFEL_E_INVALID, "activate: bad arguments");
    if (P == 0) return 0;
    launch_activate(P, theta, act, static_cast<hipStream_t>(stream));
    return launched("activate_kernel");
}

int surfel_adam_step(int P, float* theta, const float* grad, float* m, float* v, float* act, const float* lr, float beta1, float beta2,
                     float eps, int t, float grad_scale, int D, int N, const float* campos_all, const float* gcol_all, int parts, void* stream) {
    if (P < 0 || t < 1 || !lr || (P > 0 && (!theta || !grad || !m || !v || !act))) return api_fail(SURFEL_E_INVALID, "adam_step: bad arguments");
    if (gcol_all && (!campos_all || N < 1 || D < 0 || D > 3)) return api_fail(SURFEL_E_INVALID, "adam_step: bad colour-gradient arguments");
    if (parts < 1 || parts > 3) return api_fail(SURFEL_E_INVALID, "adam_step: parts must be 1 (SH block), 2 (geometry sections) or 3 (both)");
    if (P == 0) return 0;
    const float bc1 = (float)(1.0 - std::pow((double)beta1, (double)t));
    const float bc2s = (float)std::sqrt(1.0 - std::pow((double)beta2, (double)t));
    launch_adam(P, theta, grad, m, v, act, lr, beta1, beta2, eps, bc1, bc2s, grad_scale, D, N, campos_all, gcol_all, parts, static_cast<hipStream_t>(stream));
    return launched("adam kernels");
}

int surfel_train_update(int P, float* theta, const float* grad, float* m, float* v, float* act, const float* lr, float beta1, float beta2, float eps,
                        int t, float grad_scale, int D, int N, const float* campos_all, const float* gcol_all, const float* dL_dmeans2D,
                        const int* radii, float* grad_accum, float* denom, float* max_radii, void* stream) {
    if (P < 0 || t < 1 || !lr || (P > 0 && (!theta || !grad || !m || !v || !act))) return api_fail(SURFEL_E_INVALID, "train_update: bad arguments");
    if (P == 0) return 0;
    if (!gcol_all || !campos_all || N < 1 || D < 0 || D > 3) return api_fail(SURFEL_E_INVALID, "train_update: needs the colour gradients (SH block rebuilt in the kernel)");
    if (dL_dmeans2D && (!radii || !grad_accum || !denom || !max_radii)) return api_fail(SURFEL_E_INVALID, "train_update: bad statistics arguments");
    const float bc1 = (float)(1.0 - std::pow((double)beta1, (double)t));
    const float bc2s = (float)std::sqrt(1.0 - std::pow((double)beta2, (double)t));
    launch_train_update(P, theta, grad, m, v, act, lr, beta1, beta2, eps, bc1, bc2s, grad_scale, D, N, campos_all, gcol_all, dL_dmeans2D, radii, grad_accum,
                        denom, max_radii, static_cast<hipStream_t>(stream));
    return launched("train_update_kernel");
}

int surfel_sh_grad_gather(int P, int D, int N, const float* means3D, const float* campos_all, const float* gcol_all, float* dL_dsh, void* stream) {
    if (P < 0 || D < 0 || D > 3 || N < 1 || (P > 0 && (!means3D || !campos_all || !gcol_all || !dL_dsh)))
        return api_fail(SURFEL_E_INVALID, "sh_grad_gather: bad arguments");
    if (P == 0) return 0;
    launch_sh_grad_gather(P, D, N, means3D, campos_all, gcol_all, dL_dsh, static_cast<hipStream_t>(stream));
    return launched("sh_grad_gather_kernel");
}

int surfel_densify_stats(int P, const float* dL_dmeans2D, const int* radii, float* grad_accum, float* denom, float* max_radii, void* stream) {
    if (P < 0 || (P > 0 && (!dL_dmeans2D || !radii || !grad_accum || !denom || !max_radii)))
        return api_fail(SURFEL_E_INVALID, "densify_stats: bad arguments");
    if (P == 0) return 0;
    launch_densify_stats(P, dL_dmeans2D, radii, grad_accum, denom, max_radii, static_cast<hipStream_t>(stream));
    return launched("densify_stats_kernel");
}

}  // extern "C"
