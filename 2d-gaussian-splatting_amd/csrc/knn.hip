// knn.hip — simple-knn replacement for gfx950: mean squared distance to the 3 nearest neighbours.
// Replaces simple_knn._C.distCUDA2 (call site /root/reference/scene/gaussian_model.py:134; the
// CUDA submodule itself is absent, /root/reference/.gitmodules:4-6).  Exact (not approximate):
// points are Morton-sorted, grouped into boxes of 1024, and every workgroup of 256 consecutive
// sorted points scans — through LDS, with coalesced loads — only the boxes that can still hold a
// closer neighbour for at least one of its points.
#include <cstring>
#include <float.h>

#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>

#include "surfel_kernels.h"

namespace surfel {

constexpr int KNN_BOX = 1024;
constexpr int KNN_BLOCK = 256;

__device__ __forceinline__ uint32_t f2ord(float f) {
    const uint32_t b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float ord2f(uint32_t u) {
    const uint32_t b = (u & 0x80000000u) ? (u & 0x7fffffffu) : ~u;
    return __uint_as_float(b);
}

__global__ void knn_init_minmax(uint32_t* mm) {
    if (threadIdx.x < 3) mm[threadIdx.x] = 0xffffffffu;
    else if (threadIdx.x < 6) mm[threadIdx.x] = 0u;
}

__global__ void __launch_bounds__(256) knn_minmax_kernel(int P, const float* __restrict__ pts, uint32_t* mm) {
    float lo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, hi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < P; i += gridDim.x * blockDim.x) {
#pragma unroll
        for (int c = 0; c < 3; c++) { const float v = pts[3 * (size_t)i + c]; lo[c] = fminf(lo[c], v); hi[c] = fmaxf(hi[c], v); }
    }
#pragma unroll
    for (int c = 0; c < 3; c++) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { lo[c] = fminf(lo[c], __shfl_xor(lo[c], o)); hi[c] = fmaxf(hi[c], __shfl_xor(hi[c], o)); }
    }
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int c = 0; c < 3; c++) { atomicMin(&mm[c], f2ord(lo[c])); atomicMax(&mm[3 + c], f2ord(hi[c])); }
    }
}

__device__ __forceinline__ uint32_t spread10(uint32_t x) {
    x = (x | (x << 16)) & 0x030000FFu;
    x = (x | (x << 8)) & 0x0300F00Fu;
    x = (x | (x << 4)) & 0x030C30C3u;
    x = (x | (x << 2)) & 0x09249249u;
    return x;
}

__global__ void __launch_bounds__(256) knn_morton_kernel(int P, const float* __restrict__ pts, const uint32_t* __restrict__ mm,
                                                         uint32_t* __restrict__ codes, uint32_t* __restrict__ idx) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P) return;
    uint32_t code = 0;
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const float lo = ord2f(mm[c]), hi = ord2f(mm[3 + c]);
        const float ext = hi - lo;
        const float n = ext > 0.f ? (pts[3 * (size_t)i + c] - lo) / ext : 0.f;
        const uint32_t q = (uint32_t)fminf(fmaxf(n * 1023.f, 0.f), 1023.f);
        code |= spread10(q) << (2 - c);
    }
    codes[i] = code;
    idx[i] = (uint32_t)i;
}

// gather points into sorted order (float4 for 16-B loads) and compute per-box bounds
__global__ void __launch_bounds__(KNN_BLOCK) knn_gather_boxes_kernel(int P, const float* __restrict__ pts, const uint32_t* __restrict__ idx_sorted,
                                                                     float4* __restrict__ sorted, float* __restrict__ boxes) {
    __shared__ float s_lo[4][3], s_hi[4][3];
    const int b = blockIdx.x;
    float lo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, hi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    for (int k = threadIdx.x; k < KNN_BOX; k += KNN_BLOCK) {
        const int i = b * KNN_BOX + k;
        if (i < P) {
            const uint32_t src = idx_sorted[i];
            const float x = pts[3 * (size_t)src], y = pts[3 * (size_t)src + 1], z = pts[3 * (size_t)src + 2];
            sorted[i] = make_float4(x, y, z, 0.f);
            lo[0] = fminf(lo[0], x); lo[1] = fminf(lo[1], y); lo[2] = fminf(lo[2], z);
            hi[0] = fmaxf(hi[0], x); hi[1] = fmaxf(hi[1], y); hi[2] = fmaxf(hi[2], z);
        }
    }
#pragma unroll
    for (int c = 0; c < 3; c++) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { lo[c] = fminf(lo[c], __shfl_xor(lo[c], o)); hi[c] = fmaxf(hi[c], __shfl_xor(hi[c], o)); }
    }
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int c = 0; c < 3; c++) { s_lo[w][c] = lo[c]; s_hi[w][c] = hi[c]; }
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        const int c = threadIdx.x;
        boxes[6 * b + c] = fminf(fminf(s_lo[0][c], s_lo[1][c]), fminf(s_lo[2][c], s_lo[3][c]));
        boxes[6 * b + 3 + c] = fmaxf(fmaxf(s_hi[0][c], s_hi[1][c]), fmaxf(s_hi[2][c], s_hi[3][c]));
    }
}

__device__ __forceinline__ void update3(float d, float& b0, float& b1, float& b2) {
    if (d < b2) {
        if (d < b1) { b2 = b1; if (d < b0) { b1 = b0; b0 = d; } else b1 = d; }
        else b2 = d;
    }
}

__global__ void __launch_bounds__(KNN_BLOCK) knn_search_kernel(int P, int nboxes, const float4* __restrict__ sorted,
                                                               const float* __restrict__ boxes, const uint32_t* __restrict__ idx_sorted,
                                                               float* __restrict__ out) {
    __shared__ float4 s_pts[KNN_BLOCK];
    const int i = blockIdx.x * KNN_BLOCK + threadIdx.x;
    const bool valid = i < P;
    const float4 p = valid ? sorted[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    float b0 = FLT_MAX, b1 = FLT_MAX, b2 = FLT_MAX;
    // pass 0: own workgroup's points (Morton neighbours) give a tight first bound
    s_pts[threadIdx.x] = p;
    __syncthreads();
    {
        const int cnt = min(KNN_BLOCK, P - blockIdx.x * KNN_BLOCK);
        if (valid)
            for (int k = 0; k < cnt; k++) {
                if (k == (int)threadIdx.x) continue;
                const float4 q = s_pts[k];
                const float dx = p.x - q.x, dy = p.y - q.y, dz = p.z - q.z;
                update3(dx * dx + dy * dy + dz * dz, b0, b1, b2);
            }
    }
    const int own_lo = blockIdx.x * KNN_BLOCK, own_hi = own_lo + KNN_BLOCK;
    for (int b = 0; b < nboxes; b++) {
        const float lx = boxes[6 * b], ly = boxes[6 * b + 1], lz = boxes[6 * b + 2];
        const float hx = boxes[6 * b + 3], hy = boxes[6 * b + 4], hz = boxes[6 * b + 5];
        const float ddx = fmaxf(fmaxf(lx - p.x, p.x - hx), 0.f), ddy = fmaxf(fmaxf(ly - p.y, p.y - hy), 0.f), ddz = fmaxf(fmaxf(lz - p.z, p.z - hz), 0.f);
        const bool need = valid && (ddx * ddx + ddy * ddy + ddz * ddz) <= b2;
        if (!__syncthreads_or(need)) continue;
        for (int c0 = b * KNN_BOX; c0 < min(P, (b + 1) * KNN_BOX); c0 += KNN_BLOCK) {
            if (c0 >= own_lo && c0 < own_hi) continue;        // own chunk already done (uniform)
            const int k = c0 + threadIdx.x;
            __syncthreads();
            if (k < P) s_pts[threadIdx.x] = sorted[k];
            __syncthreads();
            if (need) {
                const int cnt = min(KNN_BLOCK, P - c0);
                for (int q = 0; q < cnt; q++) {
                    const float4 v = s_pts[q];
                    const float dx = p.x - v.x, dy = p.y - v.y, dz = p.z - v.z;
                    update3(dx * dx + dy * dy + dz * dz, b0, b1, b2);
                }
            }
        }
    }
    if (valid) out[idx_sorted[i]] = (b0 + b1 + b2) / 3.0f;
}

struct KnnScratch {
    uint32_t *mm, *codes_a, *codes_b, *idx_a, *idx_b; float4* sorted; float* boxes; char* sort_temp; size_t sort_bytes; size_t total;
};

static KnnScratch knn_carve(void* base, int P) {
    KnnScratch k{};
    char* b = static_cast<char*>(base);
    size_t off = 0;
    auto take = [&](size_t bytes) { off = (off + 255) / 256 * 256; char* p = b ? b + off : nullptr; off += bytes; return p; };
    const int nboxes = (P + KNN_BOX - 1) / KNN_BOX;
    k.mm = reinterpret_cast<uint32_t*>(take(6 * 4));
    k.codes_a = reinterpret_cast<uint32_t*>(take((size_t)P * 4));
    k.codes_b = reinterpret_cast<uint32_t*>(take((size_t)P * 4));
    k.idx_a = reinterpret_cast<uint32_t*>(take((size_t)P * 4));
    k.idx_b = reinterpret_cast<uint32_t*>(take((size_t)P * 4));
    k.sorted = reinterpret_cast<float4*>(take((size_t)P * 16));
    k.boxes = reinterpret_cast<float*>(take((size_t)nboxes * 6 * 4));
    size_t sb = 0;
    rocprim::double_buffer<uint32_t> kq(nullptr, nullptr), vq(nullptr, nullptr);
    (void)rocprim::radix_sort_pairs(nullptr, sb, kq, vq, (size_t)P, 0, 30, (hipStream_t) nullptr);
    k.sort_bytes = sb;
    k.sort_temp = take(sb);
    k.total = (off + 255) / 256 * 256;
    return k;
}

size_t knn_scratch_bytes(int P) { return knn_carve(nullptr, P).total; }

void launch_knn(int P, const float* points, float* out, void* scratch, size_t scratch_bytes, hipStream_t s) {
    (void)scratch_bytes;
    KnnScratch k = knn_carve(scratch, P);
    const int nboxes = (P + KNN_BOX - 1) / KNN_BOX;
    hipLaunchKernelGGL(knn_init_minmax, dim3(1), dim3(64), 0, s, k.mm);
    const int rb = min((P + 255) / 256, 2048);
    hipLaunchKernelGGL(knn_minmax_kernel, dim3(rb), dim3(256), 0, s, P, points, k.mm);
    hipLaunchKernelGGL(knn_morton_kernel, dim3((P + 255) / 256), dim3(256), 0, s, P, points, k.mm, k.codes_a, k.idx_a);
    rocprim::double_buffer<uint32_t> kq(k.codes_a, k.codes_b), vq(k.idx_a, k.idx_b);
    size_t sb = k.sort_bytes;
    (void)rocprim::radix_sort_pairs(k.sort_temp, sb, kq, vq, (size_t)P, 0, 30, s);
    hipLaunchKernelGGL(knn_gather_boxes_kernel, dim3(nboxes), dim3(KNN_BLOCK), 0, s, P, points, vq.current(), k.sorted, k.boxes);
    hipLaunchKernelGGL(knn_search_kernel, dim3((P + KNN_BLOCK - 1) / KNN_BLOCK), dim3(KNN_BLOCK), 0, s, P, nboxes, k.sorted, k.boxes,
                       vq.current(), out);
}

}  // namespace surfel
