// train_optim.hip — the surfel parameter store's kernels on gfx950: activations, fused Adam (+ activation backward
// + next iteration's activations) and the densification statistics.  Pure HBM streaming (28 B per parameter float).
//
// Store layout (raw parameters, gradients, Adam moments, all-reduce bucket all share it), P surfels:
//   xyz 3P | opacity P | scaling 2P | rotation 4P | sh 48P      = 58 floats / surfel
// (the 10 geometry floats first: under view-parallel training only they are all-reduced; the SH block is rebuilt from an
// all-gather of 3 floats / surfel / rank, see sh_grad_gather_kernel)
// Reference semantics restated: scene/gaussian_model.py:95-115 (exp / normalize / sigmoid), :153-162 (six Adam
// groups, eps 1e-15), torch.optim.Adam's update rule, train.py:126-128 + gaussian_model.py:405-407 (statistics).
#include <hip/hip_runtime.h>

#include "train_kernels.h"

namespace surfel {

namespace {

struct AdamK {
    float lr[6];          // xyz, f_dc, f_rest, opacity, scaling, rotation
    float beta1, beta2, eps;
    float bc1, bc2_sqrt;  // 1 - beta1^t, sqrt(1 - beta2^t)
    float grad_scale;
};

// torch.optim.Adam (single-tensor form): m = b1 m + (1-b1) g ; v = b2 v + (1-b2) g^2 ;
// p -= (lr / bc1) * m / (sqrt(v) / sqrt(bc2) + eps)
__device__ __forceinline__ float adam_update(float p, float g, float& m, float& v, float lr, const AdamK& k) {
    m = k.beta1 * m + (1.f - k.beta1) * g;
    v = k.beta2 * v + (1.f - k.beta2) * g * g;
    const float denom = sqrtf(v) / k.bc2_sqrt + k.eps;
    return p - (lr / k.bc1) * (m / denom);
}

__device__ __forceinline__ float sigmoidf(float x) { return 1.f / (1.f + expf(-x)); }

// elementwise sections: xyz (3P floats at offset 0) and sh (48P floats at offset 10P); j runs over the 51P of them
__global__ __launch_bounds__(256) void adam_elem_kernel(size_t n_xyz, size_t n_all, size_t sh_shift, float* __restrict__ theta,
                                                        const float* __restrict__ grad, float* __restrict__ m, float* __restrict__ v, AdamK k) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x; j < n_all; j += stride) {
        float lr;
        size_t i = j;
        if (j < n_xyz) lr = k.lr[0];
        else { lr = ((j - n_xyz) % 48) < 3 ? k.lr[1] : k.lr[2]; i = j + sh_shift; }
        float mi = m[i], vi = v[i];
        theta[i] = adam_update(theta[i], grad[i] * k.grad_scale, mi, vi, lr, k);
        m[i] = mi; v[i] = vi;
    }
}

// per-surfel sections behind activation functions; also refreshes the activated values
__device__ __forceinline__ void adam_act_surfel(int i, int P, float* __restrict__ theta, const float* __restrict__ grad, float* __restrict__ m,
                                                float* __restrict__ v, float* __restrict__ act, const AdamK& k) {
    const size_t o_op = (size_t)3 * P, o_sc = (size_t)4 * P, o_rot = (size_t)6 * P;
#pragma unroll
    for (int j = 0; j < 3; j++) {   // xyz (no activation): here rather than in a launch of its own
        const size_t o = 3 * (size_t)i + j;
        float mi = m[o], vi = v[o];
        theta[o] = adam_update(theta[o], grad[o] * k.grad_scale, mi, vi, k.lr[0], k);
        m[o] = mi; v[o] = vi;
    }
    {   // opacity = sigmoid(x)
        const size_t o = o_op + i;
        const float x = theta[o], s = sigmoidf(x);
        const float g = grad[o] * k.grad_scale * s * (1.f - s);
        float mi = m[o], vi = v[o];
        const float xn = adam_update(x, g, mi, vi, k.lr[3], k);
        theta[o] = xn; m[o] = mi; v[o] = vi;
        act[i] = sigmoidf(xn);
    }
#pragma unroll
    for (int j = 0; j < 2; j++) {   // scaling = exp(x)
        const size_t o = o_sc + 2 * (size_t)i + j;
        const float x = theta[o], s = expf(x);
        const float g = grad[o] * k.grad_scale * s;
        float mi = m[o], vi = v[o];
        const float xn = adam_update(x, g, mi, vi, k.lr[4], k);
        theta[o] = xn; m[o] = mi; v[o] = vi;
        act[(size_t)P + 2 * (size_t)i + j] = expf(xn);
    }
    {   // rotation = q / max(|q|, 1e-12)
        const size_t o = o_rot + 4 * (size_t)i;
        float q[4], g[4];
#pragma unroll
        for (int j = 0; j < 4; j++) { q[j] = theta[o + j]; g[j] = grad[o + j] * k.grad_scale; }
        const float len = sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
        const float inv = 1.f / fmaxf(len, 1e-12f);
        float gq[4];
        if (len >= 1e-12f) {
            const float ug = (q[0] * g[0] + q[1] * g[1] + q[2] * g[2] + q[3] * g[3]) * inv;
#pragma unroll
            for (int j = 0; j < 4; j++) gq[j] = (g[j] - q[j] * inv * ug) * inv;
        } else {
#pragma unroll
            for (int j = 0; j < 4; j++) gq[j] = g[j] * inv;
        }
        float qn[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            float mi = m[o + j], vi = v[o + j];
            qn[j] = adam_update(q[j], gq[j], mi, vi, k.lr[5], k);
            theta[o + j] = qn[j]; m[o + j] = mi; v[o + j] = vi;
        }
        const float ln = sqrtf(qn[0] * qn[0] + qn[1] * qn[1] + qn[2] * qn[2] + qn[3] * qn[3]);
        const float in = 1.f / fmaxf(ln, 1e-12f);
#pragma unroll
        for (int j = 0; j < 4; j++) act[(size_t)3 * P + 4 * (size_t)i + j] = qn[j] * in;
    }
}

__global__ __launch_bounds__(256) void adam_act_kernel(int P, float* __restrict__ theta, const float* __restrict__ grad, float* __restrict__ m,
                                                       float* __restrict__ v, float* __restrict__ act, AdamK k) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P) return;
    adam_act_surfel(i, P, theta, grad, m, v, act, k);
}

__global__ __launch_bounds__(256) void activate_kernel(int P, const float* __restrict__ theta, float* __restrict__ act) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P) return;
    const size_t o_op = (size_t)3 * P, o_sc = (size_t)4 * P, o_rot = (size_t)6 * P;
    act[i] = sigmoidf(theta[o_op + i]);
    act[(size_t)P + 2 * (size_t)i] = expf(theta[o_sc + 2 * (size_t)i]);
    act[(size_t)P + 2 * (size_t)i + 1] = expf(theta[o_sc + 2 * (size_t)i + 1]);
    float q[4];
#pragma unroll
    for (int j = 0; j < 4; j++) q[j] = theta[o_rot + 4 * (size_t)i + j];
    const float len = sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    const float inv = 1.f / fmaxf(len, 1e-12f);
#pragma unroll
    for (int j = 0; j < 4; j++) act[(size_t)3 * P + 4 * (size_t)i + j] = q[j] * inv;
}

__device__ __forceinline__ void densify_surfel(int i, const float* __restrict__ g2d, const int* __restrict__ radii, float* __restrict__ accum,
                                               float* __restrict__ denom, float* __restrict__ maxr) {
    const int r = radii[i];
    if (r <= 0) return;
    const float gx = g2d[3 * (size_t)i], gy = g2d[3 * (size_t)i + 1], gz = g2d[3 * (size_t)i + 2];
    accum[i] += sqrtf(gx * gx + gy * gy + gz * gz);
    denom[i] += 1.f;
    maxr[i] = fmaxf(maxr[i], (float)r);
}
__global__ __launch_bounds__(256) void densify_stats_kernel(int P, const float* __restrict__ g2d, const int* __restrict__ radii,
                                                            float* __restrict__ accum, float* __restrict__ denom, float* __restrict__ maxr) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P) return;
    densify_surfel(i, g2d, radii, accum, denom, maxr);
}

// View-parallel training: dL/dSH of the summed loss is  sum_r basis(dir(mean, campos_r)) (x) g_r  with g_r = rank r's
// clamp-masked dL/dcolour (3 floats / surfel).  Exchanging g_r (all-gather, 12 B/surfel/rank) and rebuilding the 48 SH
// gradients here replaces an all-reduce of 192 B/surfel; ranks are summed in rank order, so every replica gets the same bits.
// Basis exactly as preprocess_bwd (the reference's eval_sh, utils/sh_utils.py:57-112), zero above the active degree D.
__device__ __constant__ float GSH_C0 = 0.28209479177387814f;
__device__ __constant__ float GSH_C1 = 0.4886025119029199f;
__device__ __constant__ float GSH_C2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f, -1.0925484305920792f,
                                           0.5462742152960396f};
__device__ __constant__ float GSH_C3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f, 0.3731763325901154f,
                                           -0.4570457994644658f, 1.445305721320277f, -0.5900435899266435f};

// acc[48] += sum over ranks of basis_r (x) g_r for surfel i (rank order)
__device__ __forceinline__ void sh_grad_rebuild(int i, int P, int D, int N, float px, float py, float pz, const float* __restrict__ campos_all,
                                                const float* __restrict__ gcol_all, float (&acc)[48]) {
#pragma unroll
    for (int q = 0; q < 48; q++) acc[q] = 0.f;
    for (int r = 0; r < N; r++) {
        const float* __restrict__ g = gcol_all + ((size_t)r * P + i) * 3;
        const float gR[3] = {g[0], g[1], g[2]};
        if (gR[0] == 0.f && gR[1] == 0.f && gR[2] == 0.f) continue;      // culled on rank r (or fully clamped): contributes zeros
        const float dox = px - campos_all[3 * r], doy = py - campos_all[3 * r + 1], doz = pz - campos_all[3 * r + 2];
        const float il = rsqrtf(dox * dox + doy * doy + doz * doz);
        const float x = dox * il, y = doy * il, z = doz * il;
        float B[16];
#pragma unroll
        for (int k = 0; k < 16; k++) B[k] = 0.f;
        B[0] = GSH_C0;
        if (D > 0) {
            B[1] = -GSH_C1 * y; B[2] = GSH_C1 * z; B[3] = -GSH_C1 * x;
            if (D > 1) {
                const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
                B[4] = GSH_C2[0] * xy; B[5] = GSH_C2[1] * yz; B[6] = GSH_C2[2] * (2.f * zz - xx - yy); B[7] = GSH_C2[3] * xz;
                B[8] = GSH_C2[4] * (xx - yy);
                if (D > 2) {
                    B[9] = GSH_C3[0] * y * (3.f * xx - yy); B[10] = GSH_C3[1] * xy * z; B[11] = GSH_C3[2] * y * (4.f * zz - xx - yy);
                    B[12] = GSH_C3[3] * z * (2.f * zz - 3.f * xx - 3.f * yy); B[13] = GSH_C3[4] * x * (4.f * zz - xx - yy);
                    B[14] = GSH_C3[5] * z * (xx - yy); B[15] = GSH_C3[6] * x * (xx - 3.f * yy);
                }
            }
        }
#pragma unroll
        for (int k = 0; k < 16; k++) {
#pragma unroll
            for (int c = 0; c < 3; c++) acc[3 * k + c] += B[k] * gR[c];
        }
    }
}

__global__ __launch_bounds__(256) void sh_grad_gather_kernel(int P, int D, int N, const float* __restrict__ means3D,
                                                             const float* __restrict__ campos_all, const float* __restrict__ gcol_all,
                                                             float* __restrict__ dL_dsh) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P) return;
    float acc[48];
    sh_grad_rebuild(i, P, D, N, means3D[3 * (size_t)i], means3D[3 * (size_t)i + 1], means3D[3 * (size_t)i + 2], campos_all, gcol_all, acc);
    float4* __restrict__ out = reinterpret_cast<float4*>(dL_dsh + (size_t)i * 48);
#pragma unroll
    for (int q = 0; q < 12; q++) out[q] = make_float4(acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]);
}

// Adam on the SH block with the gradients rebuilt in registers from the colour gradients: the 192 B/surfel SH gradient never
// exists in HBM (the rasterizer's backward skips writing it, this kernel skips reading it).  Must run BEFORE the xyz update
// (the view directions use the positions the forward saw).
__global__ __launch_bounds__(256) void adam_sh_kernel(int P, int D, int N, float* __restrict__ theta, float* __restrict__ m, float* __restrict__ v,
                                                      const float* __restrict__ campos_all, const float* __restrict__ gcol_all, AdamK k) {
    // one workgroup = 64 surfels: wave 0 rebuilds their 48 SH gradients into LDS, then all 256 threads run Adam over the
    // 64 x 48 contiguous floats of theta / m / v (coalesced)
    __shared__ float s_g[64 * 49];
    const int tid = threadIdx.x;
    const int i0 = blockIdx.x * 64;
    if (tid < 64) {
        const int i = i0 + tid;
        if (i < P) {
            float acc[48];
            sh_grad_rebuild(i, P, D, N, theta[3 * (size_t)i], theta[3 * (size_t)i + 1], theta[3 * (size_t)i + 2], campos_all, gcol_all, acc);
#pragma unroll
            for (int q = 0; q < 48; q++) s_g[tid * 49 + q] = acc[q] * k.grad_scale;
        }
    }
    __syncthreads();
    const int n = min(64, P - i0) * 48;
    const size_t o = (size_t)10 * P + (size_t)48 * i0;
    for (int e = tid; e < n; e += 256) {
        const int sfl = e / 48, flat = e - sfl * 48;
        float mi = m[o + e], vi = v[o + e];
        theta[o + e] = adam_update(theta[o + e], s_g[sfl * 49 + flat], mi, vi, flat < 3 ? k.lr[1] : k.lr[2], k);
        m[o + e] = mi; v[o + e] = vi;
    }
}

// The whole update of an iteration in which nothing is rebuilt, as ONE launch (three before: densify_stats_kernel, adam_sh_kernel,
// adam_act_kernel — two dependent launch boundaries of ~2.5 us each and a 5-us kernel that is all latency).  One workgroup = 64
// surfels.  Before the barrier wave 0 rebuilds their SH gradients (reading the positions the forward saw), wave 1 adds their
// densification statistics; behind it all waves run Adam over the 64 x 48 SH floats and the last wave the per-surfel sections (xyz moves
// HERE: nobody else reads these 64 positions).  Same device functions as the three kernels: same bits
// (tests/test_gpu_train.py::test_fused_update_equals_the_three_launches).
__global__ __launch_bounds__(256) void train_update_kernel(int P, int D, int N, float* __restrict__ theta, const float* __restrict__ grad,
                                                           float* __restrict__ m, float* __restrict__ v, float* __restrict__ act,
                                                           const float* __restrict__ campos_all, const float* __restrict__ gcol_all,
                                                           const float* __restrict__ g2d, const int* __restrict__ radii, float* __restrict__ accum,
                                                           float* __restrict__ denom, float* __restrict__ maxr, AdamK k) {
    __shared__ float s_g[64 * 49];
    const int tid = threadIdx.x;
    const int i0 = blockIdx.x * 64;
    if (tid < 64) {
        const int i = i0 + tid;
        if (i < P) {
            float acc[48];
            sh_grad_rebuild(i, P, D, N, theta[3 * (size_t)i], theta[3 * (size_t)i + 1], theta[3 * (size_t)i + 2], campos_all, gcol_all, acc);
#pragma unroll
            for (int q = 0; q < 48; q++) s_g[tid * 49 + q] = acc[q] * k.grad_scale;
        }
    } else if (tid < 128 && g2d) {
        const int i = i0 + tid - 64;
        if (i < P) densify_surfel(i, g2d, radii, accum, denom, maxr);
    }
    __syncthreads();
    {   // (the loop of adam_sh_kernel, statement for statement: the compiler's contraction of adam_update depends on the loop around it)
        const int n = min(64, P - i0) * 48;
        const size_t o = (size_t)10 * P + (size_t)48 * i0;
        for (int e = tid; e < n; e += 256) {
            const int sfl = e / 48, flat = e - sfl * 48;
            float mi = m[o + e], vi = v[o + e];
            theta[o + e] = adam_update(theta[o + e], s_g[sfl * 49 + flat], mi, vi, flat < 3 ? k.lr[1] : k.lr[2], k);
            m[o + e] = mi; v[o + e] = vi;
        }
    }
    if (tid >= 192) {      // the last wave: the per-surfel sections (its SH elements are the fewest: 48 x 64 = 12 x 256)
        const int i = i0 + tid - 192;
        if (i < P) adam_act_surfel(i, P, theta, grad, m, v, act, k);
    }
}

}  // namespace

void launch_train_update(int P, float* theta, const float* grad, float* m, float* v, float* act, const float* lr, float beta1, float beta2,
                         float eps, float bc1, float bc2_sqrt, float grad_scale, int D, int N, const float* campos_all, const float* gcol_all,
                         const float* g2d, const int* radii, float* accum, float* denom, float* maxr, hipStream_t s) {
    AdamK k;
    for (int i = 0; i < 6; i++) k.lr[i] = lr[i];
    k.beta1 = beta1; k.beta2 = beta2; k.eps = eps; k.bc1 = bc1; k.bc2_sqrt = bc2_sqrt; k.grad_scale = grad_scale;
    hipLaunchKernelGGL(train_update_kernel, dim3((P + 63) / 64), dim3(256), 0, s, P, D, N, theta, grad, m, v, act, campos_all, gcol_all, g2d, radii, accum,
                       denom, maxr, k);
}

void launch_sh_grad_gather(int P, int D, int N, const float* means3D, const float* campos_all, const float* gcol_all, float* dL_dsh,
                           hipStream_t s) {
    hipLaunchKernelGGL(sh_grad_gather_kernel, dim3((P + 255) / 256), dim3(256), 0, s, P, D, N, means3D, campos_all, gcol_all, dL_dsh);
}

void launch_activate(int P, const float* theta, float* act, hipStream_t s) {
    hipLaunchKernelGGL(activate_kernel, dim3((P + 255) / 256), dim3(256), 0, s, P, theta, act);
}

void launch_adam(int P, float* theta, const float* grad, float* m, float* v, float* act, const float* lr, float beta1, float beta2, float eps,
                 float bc1, float bc2_sqrt, float grad_scale, int D, int N, const float* campos_all, const float* gcol_all, int parts,
                 hipStream_t s) {
    AdamK k;
    for (int i = 0; i < 6; i++) k.lr[i] = lr[i];
    k.beta1 = beta1; k.beta2 = beta2; k.eps = eps; k.bc1 = bc1; k.bc2_sqrt = bc2_sqrt; k.grad_scale = grad_scale;
    // SH block (parts & 1): rebuilt from the colour gradients inside adam_sh_kernel (before xyz moves), or read from grad by the
    // elementwise kernel.  Geometry sections (parts & 2): xyz + the activated sections, one per-surfel kernel.
    if (parts & 1) {
        if (gcol_all) {
            hipLaunchKernelGGL(adam_sh_kernel, dim3((P + 63) / 64), dim3(256), 0, s, P, D, N, theta, m, v, campos_all, gcol_all, k);
        } else {
            size_t nsh = (size_t)48 * P, blocks_sh = (nsh + 256 * 4 - 1) / (256 * 4);
            if (blocks_sh > 65536) blocks_sh = 65536;
            hipLaunchKernelGGL(adam_elem_kernel, dim3((unsigned)blocks_sh), dim3(256), 0, s, (size_t)0, nsh, (size_t)10 * P, theta, grad, m, v, k);
        }
    }
    if (parts & 2) hipLaunchKernelGGL(adam_act_kernel, dim3((P + 255) / 256), dim3(256), 0, s, P, theta, grad, m, v, act, k);
}

void launch_densify_stats(int P, const float* g2d, const int* radii, float* accum, float* denom, float* maxr, hipStream_t s) {
    hipLaunchKernelGGL(densify_stats_kernel, dim3((P + 255) / 256), dim3(256), 0, s, P, g2d, radii, accum, denom, maxr);
}

}  // namespace surfel
