// train_loss.hip — photometric loss of a training iteration on gfx950: L1 + SSIM forward and backward.
// Restates utils/loss_utils.py:23-24,43-73 of the reference (window_size x window_size Gaussian window, sigma 1.5, zero padding,
// C1 = 0.01^2, C2 = 0.03^2) as two LDS-tiled separable-convolution kernels instead of five grouped conv2d calls + ~20 elementwise
// kernels per direction.  One workgroup = 4 waves = a 32x32 output tile staged with a halo of the window radius (42x42 for the
// reference's window_size = 11; the kernels are templates on the radius, odd window sizes 3..15 are instantiated).
#include <hip/hip_runtime.h>

#include <cmath>

#include "surfel_common.h"
#include "train_kernels.h"
#include "train_loss_body.h"

namespace surfel {

namespace {

using namespace lossk;

template <int SR>
__global__ __launch_bounds__(256) void ssim_fwd_kernel(int H, int W, const float* __restrict__ img, const float* __restrict__ gt,
                                                       float* __restrict__ dmaps, size_t map_stride, float* __restrict__ partials, SsimWin win) {
    __shared__ __attribute__((aligned(16))) char smem[ssim_fwd_lds<SR>()];
    ssim_fwd_body<SR>(smem, blockIdx.x, gridDim.x, H, W, img, gt, dmaps, map_stride, partials, win);
}

template <int SR>
__global__ __launch_bounds__(256) void ssim_bwd_kernel(int H, int W, const float* __restrict__ img, const float* __restrict__ gt,
                                                       const float* __restrict__ dmaps, size_t map_stride, float c_l1, float c_ssim,
                                                       const float* __restrict__ g_l1_dev, const float* __restrict__ g_ssim_dev,
                                                       float* __restrict__ grad_img, SsimWin win) {
    __shared__ __attribute__((aligned(16))) char smem[ssim_bwd_lds<SR>()];
    ssim_bwd_body<SR>(smem, blockIdx.x, gridDim.x, H, W, img, gt, dmaps, map_stride, c_l1, c_ssim, g_l1_dev, g_ssim_dev, grad_img, win);
}

// Fixed-order reduction of per-workgroup partial sums: one workgroup per group.
__global__ __launch_bounds__(256) void reduce_partials_kernel(const float* __restrict__ partials, int n, int stride, float scale,
                                                              float* __restrict__ out) {
    __shared__ float red[256];
    const int g = blockIdx.x, k = blockIdx.y, tid = threadIdx.x;
    const float* p = partials + (size_t)g * n * stride + k;
    float acc = 0.f;
    for (int i = tid; i < n; i += 256) acc += p[(size_t)i * stride];
    red[tid] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) red[tid] += red[tid + s];
        __syncthreads();
    }
    if (tid == 0) out[g * stride + k] = red[0] * scale;
}

// All loss scalars of an iteration in one workgroup, fixed summation order (train.py:72-88):
// out = [Ll1, ssim, normal_err, dist, photometric, total],  photometric = (1-l)*Ll1 + l*(1-ssim),
// total = photometric + lambda_normal*normal_err + lambda_dist*dist.   pb may be NULL (no regularisers this iteration).
__global__ __launch_bounds__(1024) void loss_finalize_kernel(LossFinalize f) {
    // 1024 threads, 8-byte loads: the ~2 000 + 2 500 partial pairs of an 800x800 frame are two or three loads per thread, all in
    // flight at once (train_loss_body.h: loss_finalize_body)
    __shared__ float red[4][16];
    loss_finalize_body<1024>(red, f);
}

}  // namespace

int ssim_blocks(int H, int W) { return ((W + ST - 1) / ST) * ((H + ST - 1) / ST); }

// The window of loss_utils.py:29-31: exp(-(x - ws/2)^2 / (2 sigma^2)) evaluated in double (Python floats), stored as float32
// (torch.Tensor), divided by its float32 sum.  window_size 11 uses the table pinned against the reference's own tensor.
bool ssim_window(int window, lossk::SsimWin* w) {
    if (window < 3 || window > 15 || (window & 1) == 0) return false;
    for (int i = 0; i < 15; i++) w->w[i] = 0.f;
    if (window == 11) { for (int i = 0; i < 11; i++) w->w[i] = kG11[i]; return true; }
    float g[15], sum = 0.f;
    for (int i = 0; i < window; i++) {
        const double d = (double)(i - window / 2);
        g[i] = (float)exp(-(d * d) / (2.0 * 1.5 * 1.5));
        sum += g[i];
    }
    for (int i = 0; i < window; i++) w->w[i] = g[i] / sum;
    return true;
}

template <int SR>
static void ssim_fwd_launch(dim3 grid, hipStream_t s, int H, int W, const float* img, const float* gt, float* dmaps, size_t stride, float* partials,
                            const SsimWin& win) {
    hipLaunchKernelGGL(ssim_fwd_kernel<SR>, grid, dim3(256), 0, s, H, W, img, gt, dmaps, stride, partials, win);
}
template <int SR>
static void ssim_bwd_launch(dim3 grid, hipStream_t s, int H, int W, const float* img, const float* gt, const float* dmaps, size_t stride, float c_l1,
                            float c_ssim, const float* g_l1_dev, const float* g_ssim_dev, float* grad_img, const SsimWin& win) {
    hipLaunchKernelGGL(ssim_bwd_kernel<SR>, grid, dim3(256), 0, s, H, W, img, gt, dmaps, stride, c_l1, c_ssim, g_l1_dev, g_ssim_dev, grad_img, win);
}

bool launch_ssim_fwd(int window, int planes, int H, int W, const float* img, const float* gt, float* dmaps, float* partials, hipStream_t s) {
    SsimWin win;
    if (!ssim_window(window, &win)) return false;
    dim3 grid(ssim_blocks(H, W) * planes);
    const size_t stride = (size_t)planes * H * W;
    switch (window / 2) {
        case 1: ssim_fwd_launch<1>(grid, s, H, W, img, gt, dmaps, stride, partials, win); break;
        case 2: ssim_fwd_launch<2>(grid, s, H, W, img, gt, dmaps, stride, partials, win); break;
        case 3: ssim_fwd_launch<3>(grid, s, H, W, img, gt, dmaps, stride, partials, win); break;
        case 4: ssim_fwd_launch<4>(grid, s, H, W, img, gt, dmaps, stride, partials, win); break;
        case 5: ssim_fwd_launch<5>(grid, s, H, W, img, gt, dmaps, stride, partials, win); break;
        case 6: ssim_fwd_launch<6>(grid, s, H, W, img, gt, dmaps, stride, partials, win); break;
        default: ssim_fwd_launch<7>(grid, s, H, W, img, gt, dmaps, stride, partials, win); break;
    }
    return true;
}

bool launch_ssim_bwd(int window, int planes, int H, int W, const float* img, const float* gt, const float* dmaps, float c_l1, float c_ssim,
                     const float* g_l1_dev, const float* g_ssim_dev, float* grad_img, hipStream_t s) {
    SsimWin win;
    if (!ssim_window(window, &win)) return false;
    dim3 grid(ssim_blocks(H, W) * planes);
    const size_t stride = (size_t)planes * H * W;
    switch (window / 2) {
        case 1: ssim_bwd_launch<1>(grid, s, H, W, img, gt, dmaps, stride, c_l1, c_ssim, g_l1_dev, g_ssim_dev, grad_img, win); break;
        case 2: ssim_bwd_launch<2>(grid, s, H, W, img, gt, dmaps, stride, c_l1, c_ssim, g_l1_dev, g_ssim_dev, grad_img, win); break;
        case 3: ssim_bwd_launch<3>(grid, s, H, W, img, gt, dmaps, stride, c_l1, c_ssim, g_l1_dev, g_ssim_dev, grad_img, win); break;
        case 4: ssim_bwd_launch<4>(grid, s, H, W, img, gt, dmaps, stride, c_l1, c_ssim, g_l1_dev, g_ssim_dev, grad_img, win); break;
        case 5: ssim_bwd_launch<5>(grid, s, H, W, img, gt, dmaps, stride, c_l1, c_ssim, g_l1_dev, g_ssim_dev, grad_img, win); break;
        case 6: ssim_bwd_launch<6>(grid, s, H, W, img, gt, dmaps, stride, c_l1, c_ssim, g_l1_dev, g_ssim_dev, grad_img, win); break;
        default: ssim_bwd_launch<7>(grid, s, H, W, img, gt, dmaps, stride, c_l1, c_ssim, g_l1_dev, g_ssim_dev, grad_img, win); break;
    }
    return true;
}

void launch_reduce_partials(const float* partials, int groups, int n, int stride, float scale, float* out, hipStream_t s) {
    hipLaunchKernelGGL(reduce_partials_kernel, dim3(groups, stride), dim3(256), 0, s, partials, n, stride, scale, out);
}

void launch_loss_finalize(const float* pa, int na, float scale_a, const float* pb, int nb, float scale_b, float lambda_dssim,
                          float lambda_normal, float lambda_dist, float* out, float* total_out, hipStream_t s) {
    const LossFinalize f{pa, na, scale_a, pb, nb, scale_b, lambda_dssim, lambda_normal, lambda_dist, out, total_out};
    hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(1024), 0, s, f);
}

}  // namespace surfel
