// surfel_forward.hip — blend_fwd: per-tile front-to-back alpha blend, 10 output channels (gfx950).
// Reference interface replaced: renderCUDA of the absent diff_surfel_rasterization submodule
// (/root/reference/.gitmodules:1-3); semantics as stated in oracle/surfel_oracle.c stage 3.
#include "surfel_common.h"
#include "surfel_kernels.h"

namespace surfel {

// ---------------------------------------------------------------------------------------------
// blend_fwd: one workgroup (4 waves) per 16x16 tile.  Surfel records of the tile's sorted list are staged through
// LDS 256 at a time (20 KB, one 80-B gather per thread).  Each DPP row of 16 lanes owns a 4x4-pixel sub-tile and
// walks its own bitmask of the staged instances whose alpha>=1/255 bbox reaches that sub-tile: a wave therefore
// blends FOUR different instances per pass of the loop body, and a small surfel occupies issue slots only where
// it can contribute.  Per-pixel order is the staged (depth) order, so results equal the whole-tile walk.
// ---------------------------------------------------------------------------------------------
constexpr int MW = 8;             // 32-bit mask words per staged batch of 256
constexpr int MSTRIDE = MW + 2;   // + zero sentinel word, padded so each sub-tile's words start 8-B aligned

template <bool STATS>
__global__ void __launch_bounds__(BLOCK) blend_fwd_kernel(BlendFwdArgs a) {
    __shared__ float4 s_rec[BLOCK * 5];
    __shared__ __attribute__((aligned(8))) uint32_t s_mask[16 * MSTRIDE];    // [sub-tile][word]
    const int tile = block_tile(a.tile_map, a.map_flag, blockIdx.x, a.gx * a.gy);      // (tile_order_kernel: XCD-contiguous runs on uniform frames, longest lists first otherwise)
    if (tile < 0) return;
    const int tx = tile % a.gx, ty = tile / a.gx;
    int lx, ly, sub;
    thread_pixel(threadIdx.x, lx, ly, sub);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int pxi = tx * TILE + lx, pyi = ty * TILE + ly;
    const bool inside = pxi < a.W && pyi < a.H;
    const float pxf = (float)pxi, pyf = (float)pyi;
    const uint2 range = a.ranges[tile];
    const int n = (int)(range.y - range.x);
    const uint32_t* mrow = &s_mask[sub * MSTRIDE];

    bool done = !inside;
    float T = 1.f, C0 = 0.f, C1 = 0.f, C2 = 0.f, N0 = 0.f, N1 = 0.f, N2 = 0.f;
    float D = 0.f, M1 = 0.f, M2 = 0.f, dist = 0.f, med = 0.f;
    uint32_t last = 0, medc = 0;
    unsigned npairs = 0;      // STATS: (pixel, surfel) pairs this thread composited
    constexpr float MC1 = FAR_N / (FAR_N - NEAR_N);
    if (threadIdx.x < 16) s_mask[threadIdx.x * MSTRIDE + MW] = 0u;     // sentinel

    for (int base = 0; base < n; base += BLOCK) {
        if (__syncthreads_count(done) == BLOCK) break;
        const int m = min(BLOCK, n - base);
        unsigned ov = 0;
        if ((int)threadIdx.x < m) {
            const uint32_t id = a.point_list[range.x + base + threadIdx.x];
            const float4* __restrict__ src = reinterpret_cast<const float4*>(a.rec + (size_t)id * REC_F);
            const float4 v0 = src[0], v1 = src[1], v2 = src[2], v3 = src[3], v4 = src[4], v5 = src[5], v6 = src[6];
            s_rec[threadIdx.x * 5 + 0] = v0; s_rec[threadIdx.x * 5 + 1] = v1; s_rec[threadIdx.x * 5 + 2] = v2;
            s_rec[threadIdx.x * 5 + 3] = v3; s_rec[threadIdx.x * 5 + 4] = v4;
            ov = subtile_overlap(make_foot(v2, v5, v6), tx * TILE, ty * TILE);
        }
        // per-sub-tile bitmasks of the 64 instances this staging wave holds (words 2*wave, 2*wave+1)
#pragma unroll
        for (int s = 0; s < 16; s++) {
            const unsigned long long b = __ballot((ov >> s) & 1u);
            if (lane == 0) *reinterpret_cast<unsigned long long*>(&s_mask[s * MSTRIDE + 2 * wave]) = b;
        }
        __syncthreads();
        const int nw = (m + 31) >> 5;                 // mask words in use
        // row walk state: cur = unvisited instances of word widx, next = word widx+1 (prefetched)
        int widx = 0;
        uint32_t cur = mrow[0], next = mrow[1];
        {   // a row whose 16 pixels are all saturated skips the batch
            const unsigned long long db = __ballot(done);
            if ((((unsigned)(db >> (lane & 48))) & 0xffffu) == 0xffffu) { cur = 0u; widx = nw; }
        }
        for (;;) {
            if (!__any((cur != 0u) | (widx < nw - 1))) break;
            const bool act = cur != 0u;
            const int j = (widx << 5) + __builtin_ctz(cur | 0x80000000u);
            cur &= cur - 1u;
            const float4 q0 = s_rec[j * 5 + 0], q1 = s_rec[j * 5 + 1], q2 = s_rec[j * 5 + 2];
            // ray-splat intersection, branch-free: the one definition the backward walks re-decide with (surfel_common.h)
            Hit h;
            const bool hit = pair_hit(pxf, pyf, q0, q1, q2, h);
            const float depth = h.depth, alpha = h.alpha;
            const bool ok = act & (!done) & hit;
            if (__any(ok)) {
                if (ok) {
                    const float testT = T * (1.f - alpha);
                    if (testT < T_EPS) done = true;       // the terminating surfel is not composited
                    else {
                        if (STATS) npairs++;
                        const uint32_t contributor = base + j + 1;
                        const float4 q3 = s_rec[j * 5 + 3], q4 = s_rec[j * 5 + 4];
                        const float w = alpha * T;
                        const float mm = MC1 - (MC1 * NEAR_N) * SURFEL_RCP(depth);
                        dist += (mm * (mm * (1.f - T) - 2.f * M1) + M2) * w;
                        D += depth * w;
                        M1 += mm * w;
                        M2 += mm * mm * w;
                        if (T > 0.5f) { med = depth; medc = contributor; }
                        N0 += q3.x * w; N1 += q3.y * w; N2 += q3.z * w;
                        C0 += q3.w * w; C1 += q4.x * w; C2 += q4.y * w;
                        T = testT;
                        last = contributor;
                    }
                }
                if (__all(done)) break;
            }
            if (cur == 0u && widx < nw - 1) { widx++; cur = next; next = mrow[widx + 1]; }
        }
    }
    if (STATS) {      // stats[6] += composited pairs (the backward's stats[1] must count the same set)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) npairs += __shfl_xor(npairs, o);
        if (lane == 0) atomicAdd(&a.stats[6], (unsigned long long)npairs);
    }
    if (inside) {
        const size_t HW = (size_t)a.H * a.W;
        const size_t pix = (size_t)pyi * a.W + pxi;
        a.final_T[pix] = T; a.final_T[HW + pix] = M1; a.final_T[2 * HW + pix] = M2;
        a.n_contrib[pix] = last; a.n_contrib[HW + pix] = medc;
        a.out_color[pix] = C0 + T * a.bg[0];
        a.out_color[HW + pix] = C1 + T * a.bg[1];
        a.out_color[2 * HW + pix] = C2 + T * a.bg[2];
        a.out_others[pix] = D;
        a.out_others[HW + pix] = 1.f - T;
        a.out_others[2 * HW + pix] = N0; a.out_others[3 * HW + pix] = N1; a.out_others[4 * HW + pix] = N2;
        a.out_others[5 * HW + pix] = med;
        a.out_others[6 * HW + pix] = dist;
    }
}

void launch_blend_fwd(const BlendFwdArgs& a, hipStream_t s) {
    if (a.stats) hipLaunchKernelGGL(blend_fwd_kernel<true>, dim3(a.map_len), dim3(BLOCK), 0, s, a);
    else hipLaunchKernelGGL(blend_fwd_kernel<false>, dim3(a.map_len), dim3(BLOCK), 0, s, a);
}

}  // namespace surfel
