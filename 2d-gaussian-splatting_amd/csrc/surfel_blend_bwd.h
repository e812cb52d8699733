// surfel_blend_bwd.h — pieces shared by the blend-backward walks (surfel_backward.hip: rows / quad; surfel_backward_scan.hip: scan).
#pragma once
#include "surfel_common.h"
#include "surfel_kernels.h"

namespace surfel {

struct Pixel {
    float pxf, pyf;
    float gC0, gC1, gC2, g_depth, g_alpha, gN0, gN1, gN2, g_med, g_dist;     // upstream gradients
    float fM1, fM2, final_A;
    int last, medc;
    float T, X;      // running transmittance and the suffix sum (see pair_gradients)
};

__device__ __forceinline__ Pixel load_pixel(const BlendBwdArgs& a, int pxi, int pyi) {
    Pixel p;
    p.pxf = (float)pxi; p.pyf = (float)pyi;
    const bool inside = pxi < a.W && pyi < a.H;
    const size_t HW = (size_t)a.H * a.W;
    const size_t pix = (size_t)pyi * a.W + pxi;
    float T_final = 0.f;
    p.fM1 = 0.f; p.fM2 = 0.f; p.last = 0; p.medc = 0;
    p.gC0 = p.gC1 = p.gC2 = p.g_depth = p.g_alpha = p.gN0 = p.gN1 = p.gN2 = p.g_med = p.g_dist = 0.f;
    if (inside) {
        T_final = a.final_T[pix]; p.fM1 = a.final_T[HW + pix]; p.fM2 = a.final_T[2 * HW + pix];
        p.last = (int)a.n_contrib[pix]; p.medc = (int)a.n_contrib[HW + pix];
        p.gC0 = a.dL_dpix[pix]; p.gC1 = a.dL_dpix[HW + pix]; p.gC2 = a.dL_dpix[2 * HW + pix];
        p.g_depth = a.dL_dothers[pix]; p.g_alpha = a.dL_dothers[HW + pix];
        p.gN0 = a.dL_dothers[2 * HW + pix]; p.gN1 = a.dL_dothers[3 * HW + pix]; p.gN2 = a.dL_dothers[4 * HW + pix];
        p.g_med = a.dL_dothers[5 * HW + pix]; p.g_dist = a.dL_dothers[6 * HW + pix];
    }
    p.final_A = 1.f - T_final;
    p.T = T_final;
    p.X = T_final * __builtin_fmaf(a.bg[2], p.gC2, __builtin_fmaf(a.bg[1], p.gC1, a.bg[0] * p.gC0));     // suffix sum, seeded with the background term
    return p;
}

// This header's users are compiled with -ffp-contract=off and spell every fused multiply-add out, so the walk variants that share
// pair_gradients execute the SAME rounding sequence per (pixel, surfel) pair — that is what makes rows / quad bit-identical.
#define FMA(a, b, c) __builtin_fmaf((a), (b), (c))

// did the forward composite this pair?  (pos <= last and the forward's own tests, surfel_common.h)
__device__ __forceinline__ bool pair_hit(const Pixel& p, const float4 q0, const float4 q1, const float4 q2, int pos, Hit& h) {
    return pair_hit(p.pxf, p.pyf, q0, q1, q2, h) & (pos <= p.last);
}

// gradient-record slot of the (tile, surfel) instance whose staged record holds q4
__device__ __forceinline__ size_t grec_slot(const float4 q4, int tx, int ty) {
    const uint32_t basei = __float_as_uint(q4.z), rectbits = __float_as_uint(q4.w);
    const int x0 = rectbits & 1023, y0 = (rectbits >> 10) & 1023, rw = rectbits >> 20;
    return (size_t)basei + (size_t)((ty - y0) * rw + (tx - x0));
}

// BlendBwdArgs::scan_rule: the scan kernel and the rows kernel are both launched and every workgroup of both looks the frame's walk up
// in the word blend_fwd left (surfel_common.h: frame_walk) — the workgroups of the other kernel return at once.
__device__ __forceinline__ bool device_picks_scan(const BlendBwdArgs& a) { return a.walk_word[0] == WALK_SCAN; }

__device__ __forceinline__ int block_max(int v, int* s_max) {
    if (threadIdx.x == 0) *s_max = 0;
    __syncthreads();
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o));
    if ((threadIdx.x & 63) == 0) atomicMax(s_max, v);
    __syncthreads();
    return *s_max;
}

// Instances behind every pixel's last contributor (list positions > maxc) are never staged.  What becomes of their gradient records:
//   * frames with few instances (a.cut == NULL): a record of zeros each, so that preprocess_bwd can sum a surfel's records blindly
//     (1-2 records per surfel: any test in front of the fetch costs more than the zeros it saves);
//   * large frames (a.cut given): NO record.  A tile's list is ordered by (depth bits, surfel index), so "position <= maxc" is
//     decidable from the surfel's own key against the key of the instance at position maxc — the tile's CUT, 8 B per tile, which
//     preprocess_bwd checks before it fetches an 80-B record (crowded frames saturate early: half of their records were zeros,
//     written here and read back there).   cut = (depth bits, surfel index + 1) of the last staged instance; (0, 0): none.
__device__ __forceinline__ void finish_tail(const BlendBwdArgs& a, const uint2 range, int maxc, int tile, int tx, int ty) {
    if (a.cut) {
        if (threadIdx.x == 0) {
            uint2 c = make_uint2(0u, 0u);
            if (maxc > 0) {
                const uint32_t id = a.point_list[range.x + maxc - 1];
                c = make_uint2(__float_as_uint(a.depths[id]), id + 1u);
            }
            a.cut[tile] = c;
        }
        return;
    }
    for (int pos = maxc + 1 + (int)threadIdx.x; pos <= (int)(range.y - range.x); pos += BLOCK) {
        const uint32_t id = a.point_list[range.x + pos - 1];
        const float4 q4 = reinterpret_cast<const float4*>(a.rec + (size_t)id * REC_F)[4];
        float4* __restrict__ dst = reinterpret_cast<float4*>(a.grec + grec_slot(q4, tx, ty) * GREC_F);
        const float4 zz = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int q = 0; q < 5; q++) dst[q] = zz;
    }
}

__device__ __forceinline__ float4 add4(const float4 a, const float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }


}  // namespace surfel
