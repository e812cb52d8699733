// train_loss.hip — photometric loss of a training iteration on gfx950: L1 + SSIM forward and backward.
// Restates utils/loss_utils.py:23-24,43-73 of the reference (11x11 window, sigma 1.5, zero padding, C1 = 0.01^2,
// C2 = 0.03^2) as two LDS-tiled separable-convolution kernels instead of five grouped conv2d calls + ~20 elementwise
// kernels per direction.  One workgroup = 4 waves = a 32x32 output tile staged with a 5-pixel halo (42x42).
#include <hip/hip_runtime.h>

#include "train_kernels.h"

namespace surfel {

namespace {

constexpr int ST = 32;             // output tile edge
constexpr int SR = 5;              // window radius
constexpr int SHALO = ST + 2 * SR; // 42
constexpr int SPITCH = SHALO + 1;  // LDS row pitch (odd: the two half-waves of a ds_read_b32 land on distinct banks)
constexpr float SSIM_C1 = 0.01f * 0.01f;
constexpr float SSIM_C2 = 0.03f * 0.03f;

// gaussian(11, 1.5) of loss_utils.py:29-31 evaluated in fp32 exactly as torch does (exp in double, stored fp32, fp32 sum)
__device__ __constant__ float kG[11] = {1.028380124e-03f, 7.598758209e-03f, 3.600077331e-02f, 1.093606874e-01f,
                                        2.130055279e-01f, 2.660117149e-01f, 2.130055279e-01f, 1.093606874e-01f,
                                        3.600077331e-02f, 7.598758209e-03f, 1.028380124e-03f};

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__global__ __launch_bounds__(256) void ssim_fwd_kernel(int H, int W, const float* __restrict__ img, const float* __restrict__ gt,
                                                       float* __restrict__ dmaps, size_t map_stride, float* __restrict__ partials) {
    __shared__ float sx[SHALO * SPITCH], sy[SHALO * SPITCH];
    __shared__ float hz[5][SHALO * ST];
    __shared__ float red[8];
    const int tid = threadIdx.x;
    const int plane = blockIdx.z;
    const int x0 = blockIdx.x * ST, y0 = blockIdx.y * ST;
    const size_t poff = (size_t)plane * H * W;
    const float* X = img + poff;
    const float* Y = gt + poff;
    for (int i = tid; i < SHALO * SHALO; i += 256) {
        const int r = i / SHALO, c = i - r * SHALO;
        const int gy = y0 + r - SR, gx = x0 + c - SR;
        const bool in = gy >= 0 && gy < H && gx >= 0 && gx < W;
        sx[r * SPITCH + c] = in ? X[(size_t)gy * W + gx] : 0.f;
        sy[r * SPITCH + c] = in ? Y[(size_t)gy * W + gx] : 0.f;
    }
    __syncthreads();
    // horizontal pass: 42 rows x 32 columns, five moments
    for (int i = tid; i < SHALO * ST; i += 256) {
        const int r = i >> 5, c = i & 31;
        float a = 0.f, b = 0.f, aa = 0.f, bb = 0.f, ab = 0.f;
#pragma unroll
        for (int k = 0; k < 11; k++) {
            const float xv = sx[r * SPITCH + c + k], yv = sy[r * SPITCH + c + k], w = kG[k];
            const float wx = w * xv, wy = w * yv;
            a += wx; b += wy; aa += wx * xv; bb += wy * yv; ab += wx * yv;
        }
        hz[0][i] = a; hz[1][i] = b; hz[2][i] = aa; hz[3][i] = bb; hz[4][i] = ab;
    }
    __syncthreads();
    // vertical pass: thread -> column tid&31, rows (tid>>5) + 8*j
    const int c = tid & 31, r0 = tid >> 5;
    float l1 = 0.f, ss = 0.f;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int r = r0 + 8 * j;
        float mu1 = 0.f, mu2 = 0.f, e11 = 0.f, e22 = 0.f, e12 = 0.f;
#pragma unroll
        for (int k = 0; k < 11; k++) {
            const float w = kG[k];
            const int o = (r + k) * ST + c;
            mu1 += w * hz[0][o]; mu2 += w * hz[1][o]; e11 += w * hz[2][o]; e22 += w * hz[3][o]; e12 += w * hz[4][o];
        }
        const int gy = y0 + r, gx = x0 + c;
        if (gy < H && gx < W) {
            const float mu1_sq = mu1 * mu1, mu2_sq = mu2 * mu2, mu12 = mu1 * mu2;
            const float s1 = e11 - mu1_sq, s2 = e22 - mu2_sq, s12 = e12 - mu12;
            const float A = 2.f * mu12 + SSIM_C1, B = 2.f * s12 + SSIM_C2;
            const float C = mu1_sq + mu2_sq + SSIM_C1, D = s1 + s2 + SSIM_C2;
            const float iCD = 1.f / (C * D);
            const float S = A * B * iCD;
            const float xv = sx[(r + SR) * SPITCH + c + SR], yv = sy[(r + SR) * SPITCH + c + SR];
            l1 += fabsf(xv - yv);
            ss += S;
            if (dmaps) {
                const size_t o = poff + (size_t)gy * W + gx;
                dmaps[o] = 2.f * mu2 * (B - A) * iCD + 2.f * mu1 * S * (1.f / D - 1.f / C);   // dS/dmu1 (mu1, E[x^2], E[xy] independent)
                dmaps[map_stride + o] = -S / D;                                             // dS/dE[x^2]
                dmaps[2 * map_stride + o] = 2.f * A * iCD;                                  // dS/dE[xy]
            }
        }
    }
    l1 = wave_sum(l1); ss = wave_sum(ss);
    if ((tid & 63) == 0) { red[2 * (tid >> 6)] = l1; red[2 * (tid >> 6) + 1] = ss; }
    __syncthreads();
    if (tid == 0) {
        const size_t blk = ((size_t)plane * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
        partials[2 * blk] = (red[0] + red[2]) + (red[4] + red[6]);
        partials[2 * blk + 1] = (red[1] + red[3]) + (red[5] + red[7]);
    }
}

__global__ __launch_bounds__(256) void ssim_bwd_kernel(int H, int W, const float* __restrict__ img, const float* __restrict__ gt,
                                                       const float* __restrict__ dmaps, size_t map_stride, float c_l1, float c_ssim,
                                                       const float* __restrict__ g_l1_dev, const float* __restrict__ g_ssim_dev,
                                                       float* __restrict__ grad_img) {
    __shared__ float sm[3][SHALO * SPITCH];
    __shared__ float hz[3][SHALO * ST];
    const int tid = threadIdx.x;
    const int plane = blockIdx.z;
    const int x0 = blockIdx.x * ST, y0 = blockIdx.y * ST;
    const size_t poff = (size_t)plane * H * W;
    for (int i = tid; i < SHALO * SHALO; i += 256) {
        const int r = i / SHALO, c = i - r * SHALO;
        const int gy = y0 + r - SR, gx = x0 + c - SR;
        const bool in = gy >= 0 && gy < H && gx >= 0 && gx < W;
        const size_t o = poff + (size_t)gy * W + gx;
#pragma unroll
        for (int m = 0; m < 3; m++) sm[m][r * SPITCH + c] = in ? dmaps[m * map_stride + o] : 0.f;
    }
    __syncthreads();
    for (int i = tid; i < SHALO * ST; i += 256) {
        const int r = i >> 5, c = i & 31;
        float a = 0.f, b = 0.f, d = 0.f;
#pragma unroll
        for (int k = 0; k < 11; k++) {
            const float w = kG[k];
            a += w * sm[0][r * SPITCH + c + k]; b += w * sm[1][r * SPITCH + c + k]; d += w * sm[2][r * SPITCH + c + k];
        }
        hz[0][i] = a; hz[1][i] = b; hz[2][i] = d;
    }
    __syncthreads();
    const float k_l1 = c_l1 * (g_l1_dev ? g_l1_dev[0] : 1.f), k_ss = c_ssim * (g_ssim_dev ? g_ssim_dev[0] : 1.f);
    const int c = tid & 31, r0 = tid >> 5;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int r = r0 + 8 * j;
        const int gy = y0 + r, gx = x0 + c;
        if (gy >= H || gx >= W) continue;
        float a = 0.f, b = 0.f, d = 0.f;
#pragma unroll
        for (int k = 0; k < 11; k++) {
            const float w = kG[k];
            const int o = (r + k) * ST + c;
            a += w * hz[0][o]; b += w * hz[1][o]; d += w * hz[2][o];
        }
        const size_t o = poff + (size_t)gy * W + gx;
        const float xv = img[o], yv = gt[o];
        const float df = xv - yv;
        const float sgn = df > 0.f ? 1.f : (df < 0.f ? -1.f : 0.f);
        grad_img[o] = k_l1 * sgn + k_ss * (a + 2.f * xv * b + yv * d);
    }
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
__global__ __launch_bounds__(256) void loss_finalize_kernel(const float* __restrict__ pa, int na, float scale_a, const float* __restrict__ pb,
                                                            int nb, float scale_b, float lambda_dssim, float lambda_normal,
                                                            float lambda_dist, float* __restrict__ out) {
    __shared__ float red[4][256];
    const int tid = threadIdx.x;
    float a0 = 0.f, a1 = 0.f, b0 = 0.f, b1 = 0.f;
    for (int i = tid; i < na; i += 256) { a0 += pa[2 * (size_t)i]; a1 += pa[2 * (size_t)i + 1]; }
    if (pb) for (int i = tid; i < nb; i += 256) { b0 += pb[2 * (size_t)i]; b1 += pb[2 * (size_t)i + 1]; }
    red[0][tid] = a0; red[1][tid] = a1; red[2][tid] = b0; red[3][tid] = b1;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) {
#pragma unroll
            for (int k = 0; k < 4; k++) red[k][tid] += red[k][tid + s];
        }
        __syncthreads();
    }
    if (tid == 0) {
        const float l1 = red[0][0] * scale_a, ss = red[1][0] * scale_a, ne = red[2][0] * scale_b, di = red[3][0] * scale_b;
        const float ph = (1.f - lambda_dssim) * l1 + lambda_dssim * (1.f - ss);
        out[0] = l1; out[1] = ss; out[2] = ne; out[3] = di; out[4] = ph;
        out[5] = ph + lambda_normal * ne + lambda_dist * di;
    }
}

}  // namespace

int ssim_blocks(int H, int W) { return ((W + ST - 1) / ST) * ((H + ST - 1) / ST); }

void launch_ssim_fwd(int planes, int H, int W, const float* img, const float* gt, float* dmaps, float* partials, hipStream_t s) {
    dim3 grid((W + ST - 1) / ST, (H + ST - 1) / ST, planes);
    hipLaunchKernelGGL(ssim_fwd_kernel, grid, dim3(256), 0, s, H, W, img, gt, dmaps, (size_t)planes * H * W, partials);
}

void launch_ssim_bwd(int planes, int H, int W, const float* img, const float* gt, const float* dmaps, float c_l1, float c_ssim,
                     const float* g_l1_dev, const float* g_ssim_dev, float* grad_img, hipStream_t s) {
    dim3 grid((W + ST - 1) / ST, (H + ST - 1) / ST, planes);
    hipLaunchKernelGGL(ssim_bwd_kernel, grid, dim3(256), 0, s, H, W, img, gt, dmaps, (size_t)planes * H * W, c_l1, c_ssim, g_l1_dev,
                       g_ssim_dev, grad_img);
}

void launch_reduce_partials(const float* partials, int groups, int n, int stride, float scale, float* out, hipStream_t s) {
    hipLaunchKernelGGL(reduce_partials_kernel, dim3(groups, stride), dim3(256), 0, s, partials, n, stride, scale, out);
}

void launch_loss_finalize(const float* pa, int na, float scale_a, const float* pb, int nb, float scale_b, float lambda_dssim,
                          float lambda_normal, float lambda_dist, float* out, hipStream_t s) {
    hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(256), 0, s, pa, na, scale_a, pb, nb, scale_b, lambda_dssim, lambda_normal,
                       lambda_dist, out);
}

}  // namespace surfel
