// surfel_backward.hip — backward kernels of the gfx950 surfel rasterizer.
//   blend_bwd       : per-tile back-to-front replay; per-surfel partial gradients are reduced
//                     lane->wave with permlane-swap / DPP adds, wave->tile through LDS, and written ONCE per
//                     (tile, surfel) instance to a gradient record — no global atomics, so the
//                     result is bit-reproducible and never crosses XCD L2s with device-scope RMWs.
//                     (A per-DPP-row walk like the forward's was measured slower here: its per-row partial
//                     sums need LDS float atomics, ~45 LDS cycles each, and the kernel turns LDS-bound.)
// Semantics: oracle/surfel_oracle.c stages 4-5 (restating the absent diff-surfel-rasterization).
#include "surfel_common.h"
#include "surfel_kernels.h"

namespace surfel {

typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

// v + (v rotated right by N lanes inside each 16-lane row): fuses into one v_add_f32_dpp.
template <int CTRL>
__device__ __forceinline__ float row_ror_add(float v) {
    return v + __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), CTRL, 0xf, 0xf, true));
}
// a,b -> [a.lo+a.hi | b.lo+b.hi] over the two 32-lane halves (v_permlane32_swap + add)
__device__ __forceinline__ float fold32(float a, float b) {
    const u32x2 r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    return __uint_as_float(r.x) + __uint_as_float(r.y);
}
// x=[x0 x1 x2 x3], y=[y0 y1 y2 y3] (16-lane rows) -> [x0+x1, y0+y1, x2+x3, y2+y3] (v_permlane16_swap + add)
__device__ __forceinline__ float fold16(float x, float y) {
    const u32x2 r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(y), false, false);
    return __uint_as_float(r.x) + __uint_as_float(r.y);
}

constexpr int BS = 128;   // instances staged per outer batch (one 80-B gather per thread of waves 0-1)
constexpr int BB = 64;    // instances per accumulate/flush sub-batch
constexpr int NVP = 20;   // 18 gradient values per instance, padded to 20 for the wave reduction

// Wave-wide sum of 20 per-lane values.  Measured issue costs on gfx950 (scripts/ubench/valu_rate.hip, v_fma = 1):
// v_add_f32_dpp 1.4, v_permlane{16,32}_swap 3.0 — so the lanes are folded INSIDE their 16-lane rows first, where
// DPP adds can merge two registers into one by writing disjoint lane banks of one destination (bank_mask), and
// only the 5 surviving registers cross rows through permlane swaps:
//   fold8 x10 (20 DPP) -> fold4 x5 (10 DPP) -> quad sum x5 (10 DPP) -> fold16 x3, fold32 x2 (5 swaps + 5 adds)
// = 40 DPP adds + 5 swaps + 5 adds (~76 issue units; folding across rows first costs 15 swaps, ~100 units).
// Result: u0 holds in quad q (lanes 4q..4q+3) of row r the total of value 4r + {0,2,1,3}[q]; u1 holds in quad q
// of row 0 the total of value 16 + {0,2,1,3}[q].
//   fold8 : out = [lanes 0-7 : a(l)+a(l+8) | lanes 8-15 : b(l)+b(l-8)]
//   fold4 : out = [bank0 : c(l)+c(l+4) | bank1 : d(l)+d(l-4) | bank2 : c | bank3 : d]
// The s_nop covers the 2 wait states a DPP source needs after a VALU write (the compiler cannot see into the asm).
__device__ __forceinline__ float row_fold8(float a, float b) {
    float out;
    asm volatile("s_nop 1\n\t"
                 "v_add_f32_dpp %0, %1, %1 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_f32_dpp %0, %2, %2 row_ror:8 row_mask:0xf bank_mask:0xc"
                 : "=&v"(out) : "v"(a), "v"(b));
    return out;
}
__device__ __forceinline__ float row_fold4(float c, float d) {
    float out;
    asm volatile("s_nop 1\n\t"
                 "v_add_f32_dpp %0, %1, %1 row_ror:12 row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_f32_dpp %0, %2, %2 row_ror:4 row_mask:0xf bank_mask:0xa"
                 : "=&v"(out) : "v"(c), "v"(d));
    return out;
}
__device__ __forceinline__ void wave_reduce20(const float (&v)[NVP], float& u0, float& u1) {
    float a[10], z[5];
#pragma unroll
    for (int i = 0; i < 10; i++) a[i] = row_fold8(v[2 * i], v[2 * i + 1]);
#pragma unroll
    for (int m = 0; m < 5; m++) {
        float t = row_fold4(a[2 * m], a[2 * m + 1]);
        t = row_ror_add<0xB1>(t);      // quad_perm [1,0,3,2]
        t = row_ror_add<0x4E>(t);      // quad_perm [2,3,0,1]
        z[m] = t;
    }
    const float t0 = fold16(z[0], z[1]), t1 = fold16(z[2], z[3]), t2 = fold16(z[4], z[4]);
    u0 = fold32(t0, t1);
    u1 = fold32(t2, t2);
}

// ---------------------------------------------------------------------------------------------
// blend_bwd
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(BLOCK) blend_bwd_kernel(BlendBwdArgs a) {
    __shared__ float4 s_rec[BS * 5];                 // 20 KB
    __shared__ float s_acc[4][BB][NVP];              // 20 KB: per-wave partial sums of the current sub-batch
    __shared__ unsigned long long s_mask[4];         // which sub-batch slots each wave wrote
    __shared__ unsigned long long s_qmask[4][4];     // [quad][staging wave] overlap bitmasks of the staged batch
    __shared__ int s_max;
    const int tile = xcd_tile(blockIdx.x, a.gx * a.gy);
    const int tx = tile % a.gx, ty = tile / a.gx;
    int lx, ly, sub;
    thread_pixel(threadIdx.x, lx, ly, sub);
    (void)sub;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int pxi = tx * TILE + lx, pyi = ty * TILE + ly;
    const bool inside = pxi < a.W && pyi < a.H;
    const float pxf = (float)pxi, pyf = (float)pyi;
    const uint2 range = a.ranges[tile];
    const size_t HW = (size_t)a.H * a.W;
    const size_t pix = (size_t)pyi * a.W + pxi;

    float T_final = 0.f, fM1 = 0.f, fM2 = 0.f;
    int last = 0, medc = 0;
    float gC0 = 0.f, gC1 = 0.f, gC2 = 0.f, g_depth = 0.f, g_alpha = 0.f, gN0 = 0.f, gN1 = 0.f, gN2 = 0.f, g_med = 0.f, g_dist = 0.f;
    if (inside) {
        T_final = a.final_T[pix]; fM1 = a.final_T[HW + pix]; fM2 = a.final_T[2 * HW + pix];
        last = (int)a.n_contrib[pix]; medc = (int)a.n_contrib[HW + pix];
        gC0 = a.dL_dpix[pix]; gC1 = a.dL_dpix[HW + pix]; gC2 = a.dL_dpix[2 * HW + pix];
        g_depth = a.dL_dothers[pix]; g_alpha = a.dL_dothers[HW + pix];
        gN0 = a.dL_dothers[2 * HW + pix]; gN1 = a.dL_dothers[3 * HW + pix]; gN2 = a.dL_dothers[4 * HW + pix];
        g_med = a.dL_dothers[5 * HW + pix]; g_dist = a.dL_dothers[6 * HW + pix];
    }
    const float final_A = 1.f - T_final;

    if (threadIdx.x == 0) s_max = 0;
    __syncthreads();
    {
        int m = last;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m = max(m, __shfl_xor(m, o));
        if (lane == 0) atomicMax(&s_max, m);
    }
    __syncthreads();
    const int maxc = s_max;

    float T = T_final;
    float X = T_final * (a.bg[0] * gC0 + a.bg[1] * gC1 + a.bg[2] * gC2);     // suffix sum, seeded with the background term
    constexpr float MC1 = FAR_N / (FAR_N - NEAR_N), MC2 = (FAR_N * NEAR_N) / (FAR_N - NEAR_N);
    // after wave_reduce20 the quad leaders hold the totals: u0 -> value 4*row + {0,2,1,3}[quad], u1 (row 0) -> 16 + ...
    const int row = lane >> 4, quad = (lane >> 2) & 3;
    const int vslot = 4 * row + (((quad & 1) << 1) | (quad >> 1));      // value index this lane's u0 holds
    const bool writer = (lane & 3) == 0;                                 // one lane per quad

    for (int hi = maxc; hi > 0; hi -= BS) {
        const int mb = min(BS, hi);
        __syncthreads();                      // previous batch fully flushed
        {
            unsigned ov = 0;
            if ((int)threadIdx.x < mb) {
                const uint32_t id = a.point_list[range.x + (hi - threadIdx.x) - 1];
                const float4* __restrict__ src = reinterpret_cast<const float4*>(a.rec + (size_t)id * REC_F);
                const float4 v0 = src[0], v1 = src[1], v2 = src[2], v3 = src[3], v4 = src[4], v5 = src[5], v6 = src[6];
                s_rec[threadIdx.x * 5 + 0] = v0; s_rec[threadIdx.x * 5 + 1] = v1; s_rec[threadIdx.x * 5 + 2] = v2;
                s_rec[threadIdx.x * 5 + 3] = v3; s_rec[threadIdx.x * 5 + 4] = v4;
                ov = quad_overlap(make_foot(v2, v5, v6), tx * TILE, ty * TILE);
            }
            const unsigned long long b0 = __ballot(ov & 1u), b1 = __ballot(ov & 2u), b2 = __ballot(ov & 4u), b3 = __ballot(ov & 8u);
            if (lane == 0) { s_qmask[0][wave] = b0; s_qmask[1][wave] = b1; s_qmask[2][wave] = b2; s_qmask[3][wave] = b3; }
        }
        __syncthreads();
        for (int sb = 0; sb * BB < mb; sb++) {        // sub-batch sb = staged instances [64 sb, 64 sb + 64)
            unsigned long long wmask = 0ull;
            unsigned long long qm = uniform_u64(s_qmask[wave][sb]);
            while (qm) {
                const int jj = __builtin_ctzll(qm);
                qm &= qm - 1;
                const int j = sb * BB + jj;
                const int pos = hi - j;           // 1-based position in the tile's list
                const float4 q0 = s_rec[j * 5 + 0], q1 = s_rec[j * 5 + 1], q2 = s_rec[j * 5 + 2];
                // ---- ray-splat intersection, branch-free
                const float Tux = q0.x, Tuy = q0.y, Tuz = q0.z, Tvx = q0.w, Tvy = q1.x, Tvz = q1.y;
                const float Twx = q1.z, Twy = q1.w, Twz = q2.x, opa = q2.w;
                const float kx = pxf * Twx - Tux, ky = pxf * Twy - Tuy, kz = pxf * Twz - Tuz;
                const float lx_ = pyf * Twx - Tvx, ly_ = pyf * Twy - Tvy, lz_ = pyf * Twz - Tvz;
                const float p0 = ky * lz_ - kz * ly_, p1 = kz * lx_ - kx * lz_, p2 = kx * ly_ - ky * lx_;
                const float ip = __builtin_amdgcn_rcpf(p2);
                const float sx = p0 * ip, sy = p1 * ip;
                const float rho3d = sx * sx + sy * sy;
                const float dx = q2.y - pxf, dy = q2.z - pyf;
                const float rho2d = FILTER_INV_SQUARE * (dx * dx + dy * dy);
                const bool use3d = rho3d <= rho2d;
                const float rho = fminf(rho3d, rho2d);
                const float depth = use3d ? (sx * Twx + sy * Twy) + Twz : Twz;
                const float G = __expf(-0.5f * rho);
                const float alpha = fminf(ALPHA_MAX, opa * G);
                const bool ok = (pos <= last) & (p2 != 0.f) & (depth >= NEAR_N) & (alpha >= ALPHA_MIN);
                if (__ballot(ok) == 0ull) continue;      // wave-uniform
                // Sequential per-pixel state.  Every upstream gradient enters dL/dalpha only through its dot product
                // with this surfel's attributes, so the back-to-front recurrences (colour, depth, alpha, normal,
                // distortion, background) collapse into ONE scalar suffix sum
                //     X_k = T_final*bg.gC + sum_{i>k} w_i u_i ,  u_i = c_i.gC + d_i g_D + g_A + n_i.gN + dDist/dw_i
                //     dL/dalpha_k = T_k u_k - X_k / (1 - alpha_k)
                // (algebraically identical to the per-channel "accum_rec" recurrences, 2 state floats instead of 17).
                const float4 q3 = s_rec[j * 5 + 3], q4 = s_rec[j * 5 + 4];
                float w = 0.f, dL_dalpha = 0.f, dL_dz = 0.f;
                if (ok) {
                    const float i1a = __builtin_amdgcn_rcpf(1.f - alpha);
                    T = T * i1a;
                    w = alpha * T;
                    const float inv_d = __builtin_amdgcn_rcpf(depth);
                    const float mm = MC1 - (MC1 * NEAR_N) * inv_d;
                    float u = (fM2 + mm * (mm * final_A - 2.f * fM1)) * g_dist + g_alpha;
                    u += q3.w * gC0; u += q4.x * gC1; u += q4.y * gC2;
                    u += depth * g_depth;
                    u += q3.x * gN0; u += q3.y * gN1; u += q3.z * gN2;
                    dL_dalpha = T * u - X * i1a;
                    X += w * u;
                    dL_dz = (2.f * w * g_dist) * (mm * final_A - fM1) * (MC2 * inv_d * inv_d) + w * g_depth;
                    dL_dz += (pos == medc) ? g_med : 0.f;
                }
                float gv[NVP];
                gv[15] = w * gC0; gv[16] = w * gC1; gv[17] = w * gC2;
                gv[11] = w * gN0; gv[12] = w * gN1; gv[13] = w * gN2;
                gv[14] = G * dL_dalpha;
                gv[18] = 0.f; gv[19] = 0.f;
                {
                    const float nGG = -G * (opa * dL_dalpha);      // dL/dG * dG/drho*2 ; 0.99 clamp is pass-through
                    // low-pass branch: no gradient reaches the intersection.  The selects zero (s, 1/p2) themselves —
                    // they may be inf there (p2 ~ 0 on edge-on discs) and 0 * inf must not enter the sums.
                    const float sxg = use3d ? sx : 0.f, syg = use3d ? sy : 0.f, ipg = use3d ? ip : 0.f;
                    const float g2 = use3d ? 0.f : nGG * FILTER_INV_SQUARE;
                    const float ax = (nGG * sxg + dL_dz * Twx) * ipg, ay = (nGG * syg + dL_dz * Twy) * ipg;
                    const float dp2 = -(ax * sxg + ay * syg);
                    // -dk = dp x l ,  -dl = k x dp
                    const float nk0 = ay * lz_ - dp2 * ly_, nk1 = dp2 * lx_ - ax * lz_, nk2 = ax * ly_ - ay * lx_;
                    const float nl0 = ky * dp2 - kz * ay, nl1 = kz * ax - kx * dp2, nl2 = kx * ay - ky * ax;
                    gv[0] = nk0; gv[1] = nk1; gv[2] = nk2;
                    gv[3] = nl0; gv[4] = nl1; gv[5] = nl2;
                    gv[6] = dL_dz * sxg - (pxf * nk0 + pyf * nl0);
                    gv[7] = dL_dz * syg - (pxf * nk1 + pyf * nl1);
                    gv[8] = dL_dz - (pxf * nk2 + pyf * nl2);
                    gv[9] = g2 * dx; gv[10] = g2 * dy;
                }
                float u0, u1;
                wave_reduce20(gv, u0, u1);
                if (writer) {
                    s_acc[wave][jj][vslot] = u0;
                    if (row == 0) s_acc[wave][jj][16 + vslot] = u1;      // row 0: values 16, 17 and the two zero pads
                }
                wmask |= 1ull << jj;
            }
            if (lane == 0) s_mask[wave] = wmask;
            __syncthreads();
            // flush: one thread per sub-batch instance sums the four wave partials in fixed order and writes the
            // instance's gradient record — always (zeros when nobody touched it), so grec needs no memset
            if ((int)threadIdx.x < min(BB, mb - sb * BB)) {
                const int jj = threadIdx.x;
                const unsigned long long bit = 1ull << jj;
                const bool h0 = s_mask[0] & bit, h1 = s_mask[1] & bit, h2 = s_mask[2] & bit, h3 = s_mask[3] & bit;
                float4 out[5];
#pragma unroll
                for (int q = 0; q < 5; q++) {
                    float4 sacc = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (h0) { const float4 t4 = *reinterpret_cast<const float4*>(&s_acc[0][jj][4 * q]); sacc.x += t4.x; sacc.y += t4.y; sacc.z += t4.z; sacc.w += t4.w; }
                    if (h1) { const float4 t4 = *reinterpret_cast<const float4*>(&s_acc[1][jj][4 * q]); sacc.x += t4.x; sacc.y += t4.y; sacc.z += t4.z; sacc.w += t4.w; }
                    if (h2) { const float4 t4 = *reinterpret_cast<const float4*>(&s_acc[2][jj][4 * q]); sacc.x += t4.x; sacc.y += t4.y; sacc.z += t4.z; sacc.w += t4.w; }
                    if (h3) { const float4 t4 = *reinterpret_cast<const float4*>(&s_acc[3][jj][4 * q]); sacc.x += t4.x; sacc.y += t4.y; sacc.z += t4.z; sacc.w += t4.w; }
                    out[q] = sacc;
                }
                const float4 q4 = s_rec[(sb * BB + jj) * 5 + 4];
                const uint32_t basei = __float_as_uint(q4.z), rectbits = __float_as_uint(q4.w);
                const int x0 = rectbits & 1023, y0 = (rectbits >> 10) & 1023, rw = rectbits >> 20;
                const size_t dest = (size_t)basei + (size_t)((ty - y0) * rw + (tx - x0));
                float4* __restrict__ dst = reinterpret_cast<float4*>(a.grec + dest * GREC_F);
#pragma unroll
                for (int q = 0; q < 5; q++) dst[q] = out[q];
            }
            __syncthreads();                  // s_acc / s_mask reusable
        }
    }
    // instances behind every pixel's last contributor were never staged: their records are zero
    for (int pos = maxc + 1 + (int)threadIdx.x; pos <= (int)(range.y - range.x); pos += BLOCK) {
        const uint32_t id = a.point_list[range.x + pos - 1];
        const float4 q4 = reinterpret_cast<const float4*>(a.rec + (size_t)id * REC_F)[4];
        const uint32_t basei = __float_as_uint(q4.z), rectbits = __float_as_uint(q4.w);
        const int x0 = rectbits & 1023, y0 = (rectbits >> 10) & 1023, rw = rectbits >> 20;
        const size_t dest = (size_t)basei + (size_t)((ty - y0) * rw + (tx - x0));
        float4* __restrict__ dst = reinterpret_cast<float4*>(a.grec + dest * GREC_F);
        const float4 zz = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int q = 0; q < 5; q++) dst[q] = zz;
    }
}

void launch_blend_bwd(const BlendBwdArgs& a, hipStream_t s) {
    hipLaunchKernelGGL(blend_bwd_kernel, dim3(a.gx * a.gy), dim3(BLOCK), 0, s, a);
}

}  // namespace surfel
