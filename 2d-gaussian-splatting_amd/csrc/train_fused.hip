// train_fused.hip — the training loss of an iteration as TWO launches instead of four (gfx950).
// The photometric half (L1 + SSIM on the image) and the geometric half (allmap post-processing + normal / distortion regularisers)
// of train.py:72-88 read different inputs and share no data, and each is a short, latency-bound kernel (26 + 9 us forward,
// 19 + 13 us backward at 800x800) — so they are fused HORIZONTALLY: one grid whose first workgroups run the SSIM body and whose last
// workgroups run the post-processing body over the same LDS workspace (max, not sum, of the two), and the device overlaps what
// used to be two dependent launches.  The bodies are the ones of train_loss.hip / train_post.hip (train_*_body.h): same bits.
#include <hip/hip_runtime.h>

#include "surfel_common.h"
#include "train_kernels.h"
#include "train_loss_body.h"
#include "train_post_body.h"

namespace surfel {

bool ssim_window(int window, lossk::SsimWin* w);      // train_loss.hip

namespace {

constexpr int SR11 = 5;      // the reference's window_size = 11 (loss_utils.py:43)
#ifndef LOSS_PAD_LDS
#define LOSS_PAD_LDS 0       // diagnostic builds: extra LDS bytes per workgroup (occupancy experiments, python build.py --variant)
#endif
constexpr size_t cmax(size_t a, size_t b) { return a > b ? a : b; }

// grid = [n_ssim SSIM workgroups, padded to a multiple of 8 so that the post-processing part keeps its XCD mapping | n_post workgroups]
__global__ __launch_bounds__(256) void train_loss_fwd_kernel(int n_ssim, int n_ssim_pad, int n_post, int H, int W, const float* __restrict__ img,
                                                             const float* __restrict__ gt, float* __restrict__ dmaps, size_t map_stride,
                                                             float* __restrict__ partials, lossk::SsimWin win, const float* __restrict__ allmap,
                                                             const float* __restrict__ cam, float ratio, float* __restrict__ maps,
                                                             float* __restrict__ post_partials) {
    __shared__ __attribute__((aligned(16))) char smem[cmax(lossk::ssim_fwd_lds<SR11>(), postk::post_fwd_lds()) + LOSS_PAD_LDS];
    const int b = blockIdx.x;
    if (b < n_ssim_pad) {
        if (b < n_ssim) lossk::ssim_fwd_body<SR11>(smem, b, n_ssim, H, W, img, gt, dmaps, map_stride, partials, win);
    } else {
        postk::post_fwd_body(smem, b - n_ssim_pad, n_post, H, W, allmap, cam, ratio, maps, post_partials);
    }
}

__global__ __launch_bounds__(256) void train_loss_bwd_kernel(int n_ssim, int n_ssim_pad, int n_post, int H, int W, const float* __restrict__ img,
                                                             const float* __restrict__ gt, const float* __restrict__ dmaps, size_t map_stride,
                                                             float c_l1, float c_ssim, const float* __restrict__ g_dev, float* __restrict__ grad_img,
                                                             lossk::SsimWin win, const float* __restrict__ allmap, const float* __restrict__ cam,
                                                             float ratio, const float* __restrict__ gmaps, float c_normal, float c_dist,
                                                             float* __restrict__ gall, lossk::LossFinalize fin) {
    __shared__ __attribute__((aligned(16))) char smem[cmax(lossk::ssim_bwd_lds<SR11>(), postk::post_bwd_lds()) + LOSS_PAD_LDS];
    int b = blockIdx.x;
    if (fin.out) {      // deferred loss scalars: the grid starts with 8 extra workgroups (8: the others keep their XCD, b % 8); the first one
        if (b == 0) lossk::loss_finalize_body<256>(reinterpret_cast<float(*)[16]>(smem), fin);      // reduces the forward's partial sums —
        if (b < 8) return;                                                                         // final since that earlier launch — while
        b -= 8;                                                                                    // the rest of the grid does its work
    }                                                                                              // (as the LAST workgroup it ran alone
    if (b < n_ssim_pad) {                                                                          // behind everything: +10 us on the kernel)
        if (b < n_ssim) lossk::ssim_bwd_body<SR11>(smem, b, n_ssim, H, W, img, gt, dmaps, map_stride, c_l1, c_ssim, g_dev, g_dev, grad_img, win);
    } else {
        postk::post_bwd_body(smem, b - n_ssim_pad, n_post, H, W, allmap, cam, ratio, gmaps, c_normal, c_dist, g_dev, gall);
    }
}

}  // namespace

void launch_train_loss_fwd(int H, int W, const float* img, const float* gt, float* dmaps, float* partials, const float* allmap, const float* cam,
                           float ratio, float* post_partials, hipStream_t s) {
    lossk::SsimWin win;
    (void)ssim_window(11, &win);
    const int n_ssim = ssim_blocks(H, W) * 3, n_pad = (n_ssim + 7) / 8 * 8, n_post = post_blocks(H, W);
    hipLaunchKernelGGL(train_loss_fwd_kernel, dim3(n_pad + n_post), dim3(256), 0, s, n_ssim, n_pad, n_post, H, W, img, gt, dmaps, (size_t)3 * H * W,
                       partials, win, allmap, cam, ratio, (float*)nullptr, post_partials);      // (maps / gmaps stay RUNTIME arguments: as literals
                                                                                                 // they would be folded into the body and could change
                                                                                                 // its floating-point contraction, i.e. its bits)
}

void launch_train_loss_bwd(int H, int W, const float* img, const float* gt, const float* dmaps, float c_l1, float c_ssim, const float* g_dev,
                           float* grad_img, const float* allmap, const float* cam, float ratio, float c_normal, float c_dist, float* gall,
                           const float* ssim_partials, const float* post_partials, float lambda_dssim, float lambda_normal, float lambda_dist,
                           float* out6, float* total_out, hipStream_t s) {
    lossk::SsimWin win;
    (void)ssim_window(11, &win);
    const int n_ssim = ssim_blocks(H, W) * 3, n_pad = (n_ssim + 7) / 8 * 8, n_post = post_blocks(H, W);
    const lossk::LossFinalize fin{ssim_partials, n_ssim, 1.f / (float)((size_t)3 * H * W), post_partials, n_post, 1.f / (float)((size_t)H * W),
                                  lambda_dssim, lambda_normal, lambda_dist, out6, total_out};
    hipLaunchKernelGGL(train_loss_bwd_kernel, dim3(n_pad + n_post + (out6 ? 8 : 0)), dim3(256), 0, s, n_ssim, n_pad, n_post, H, W, img, gt, dmaps,
                       (size_t)3 * H * W, c_l1, c_ssim, g_dev, grad_img, win, allmap, cam, ratio, (const float*)nullptr, c_normal, c_dist, gall, fin);
}

}  // namespace surfel
