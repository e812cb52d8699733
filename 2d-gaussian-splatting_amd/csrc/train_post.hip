// train_post.hip — everything render() does to the rasterizer's allmap, as one forward and one backward kernel on gfx950.
// Restates gaussian_renderer/__init__.py:118-147 (alpha / world normals / median + expected depth / surf_depth) and
// utils/point_utils.py:9-37 (back-projection + finite-difference normals) of the reference, and in fused mode adds the
// normal-consistency and distortion regularisers of train.py:80-85, so a training iteration needs two launches here
// instead of ~40 small PyTorch kernels and several full-image temporaries.
// One workgroup = a 16x16 pixel tile; points are staged in LDS with a 1-pixel (forward) / 2-pixel (backward) halo.
#include <hip/hip_runtime.h>

#include <cfloat>

#include "surfel_common.h"
#include "train_kernels.h"
#include "train_post_body.h"

namespace surfel {

namespace {

using namespace postk;

__global__ __launch_bounds__(256) void post_fwd_kernel(int H, int W, const float* __restrict__ allmap, const float* __restrict__ cam,
                                                       float ratio, float* __restrict__ maps, float* __restrict__ partials) {
    __shared__ __attribute__((aligned(16))) char smem[post_fwd_lds()];
    post_fwd_body(smem, blockIdx.x, gridDim.x, H, W, allmap, cam, ratio, maps, partials);
}

__global__ __launch_bounds__(256) void post_bwd_kernel(int H, int W, const float* __restrict__ allmap, const float* __restrict__ cam,
                                                       float ratio, const float* __restrict__ gmaps, float c_normal, float c_dist,
                                                       const float* __restrict__ gscale_dev, float* __restrict__ gall) {
    __shared__ __attribute__((aligned(16))) char smem[post_bwd_lds()];
    post_bwd_body(smem, blockIdx.x, gridDim.x, H, W, allmap, cam, ratio, gmaps, c_normal, c_dist, gscale_dev, gall);
}

}  // namespace

int post_blocks(int H, int W) { return ((W + PT - 1) / PT) * ((H + PT - 1) / PT); }

void launch_post_fwd(int H, int W, const float* allmap, const float* cam, float ratio, float* maps, float* partials, hipStream_t s) {
    dim3 grid(post_blocks(H, W));
    hipLaunchKernelGGL(post_fwd_kernel, grid, dim3(256), 0, s, H, W, allmap, cam, ratio, maps, partials);
}

void launch_post_bwd(int H, int W, const float* allmap, const float* cam, float ratio, const float* gmaps, float c_normal, float c_dist,
                     const float* gscale_dev, float* gall, hipStream_t s) {
    dim3 grid(post_blocks(H, W));
    hipLaunchKernelGGL(post_bwd_kernel, grid, dim3(256), 0, s, H, W, allmap, cam, ratio, gmaps, c_normal, c_dist, gscale_dev, gall);
}

}  // namespace surfel
