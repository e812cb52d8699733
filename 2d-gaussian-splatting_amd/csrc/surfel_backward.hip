// surfel_backward.hip — blend_bwd: per-tile back-to-front replay of the sorted instance list (gfx950).
//
// Per-surfel partial gradients are reduced with DPP adds inside 16-lane rows (+ permlane swaps in the per-wave variant),
// gathered per tile through LDS in a FIXED order and written ONCE per (tile, surfel) instance to an 80-B gradient record —
// no global atomics, so the result is bit-reproducible and never crosses XCD L2s with device-scope RMWs.
//
// Two variants, bit-identical by construction (same per-pair arithmetic, same summation tree), selected by BlendBwdArgs.variant:
//   rows (default): every DPP row of 16 lanes owns a 4x4-pixel sub-tile and walks ITS OWN instance bitmask, so a wave works on
//                   four instances per visit and a small surfel occupies issue slots only in the sub-tiles its alpha >= 1/255
//                   footprint reaches.  Row totals (18 values, 38 DPP adds in one block, no cross-row traffic) go to private LDS slots —
//                   one per (instance, overlapped sub-tile), packed by a prefix sum, 256 slots per round — and a flush pass
//                   adds an instance's slots in the tree order ((r0+r1)+(r2+r3)) per wave, waves 0..3.
//   quad          : a wave (8x8 pixels) walks the instances its quad overlaps, one per visit, wave-wide reduction
//                   (38 DPP adds + 5 permlane swaps), per-wave LDS partials.  Round 1's kernel; the faster walk on wide footprints.
// Which of the two runs is the caller's choice (BlendBwdArgs::variant); rows is the product's walk, quad the reference it is held to.
// A third walk with another structure (lanes = instances, DPP row scans; not bit-identical to these two) lives in
// surfel_backward_scan.hip.  What happens to the instances behind a tile's saturation point: surfel_blend_bwd.h (finish_tail).
// Semantics: oracle/surfel_oracle.c stages 4-5 (restating the absent diff-surfel-rasterization).
#include "surfel_blend_bwd.h"

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

constexpr int BS = 128;   // instances staged per outer batch (one 112-B gather per thread of waves 0-1)
constexpr int BB = 64;    // quad variant: instances per accumulate/flush sub-batch
constexpr int NVP = 20;   // 18 gradient values per instance, padded to 20 for the reductions
#ifndef ROWS_RSHIFT
#define ROWS_RSHIFT 8
#endif
#ifndef ROWS_WAVES
#define ROWS_WAVES 4      // 128 VGPRs (the stream pieces of the next batch ride across the flush); 4 -> 5 workgroups per CU measured 0 % (profiles/r02_blend_bwd_variants.md)
#endif
constexpr int RSHIFT = ROWS_RSHIFT;
constexpr int RSLOTS = 1 << RSHIFT;   // rows variant: (instance, sub-tile) slots per round
constexpr int NSLOT = RSLOTS + 15;    // the last instance of a round may run 15 slots past the cut (271 slots: LDS stays <= 32 KB -> 5 workgroups / CU)

// Sum of 20 per-lane values over each 16-lane row.  Measured issue costs on gfx950 (scripts/ubench/valu_rate.hip, v_fma = 1):
// v_add_f32_dpp 1.4, v_permlane{16,32}_swap 3.0 — so lanes are folded INSIDE their rows, where DPP adds can merge two
// registers into one by writing disjoint lane banks of one destination (bank_mask):
//   fold8 x10 (20 DPP) -> fold4 x5 (10 DPP) -> quad sum x5 (10 DPP)  = 40 DPP adds
// Result: z[m] holds, in every lane of quad q (lanes 4q..4q+3) of a row, that row's total of value 4m + {0,2,1,3}[q].
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
// sum over the 4 lanes of every quad, result in all 4 lanes (two fused DPP adds)
__device__ __forceinline__ float quad_sum(float t) {
    float u, out;
    asm volatile("s_nop 1\n\t"
                 "v_add_f32_dpp %0, %2, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\t"
                 "v_add_f32_dpp %1, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf"
                 : "=&v"(u), "=&v"(out) : "v"(t));
    return out;
}
__device__ __forceinline__ void row_reduce20(const float (&v)[NVP], float (&z)[5]) {
    // One block, all in place: value 2i folds value 2i+1 into its upper lanes, then 4m folds 4m+2, then the quads are summed.
    // Every DPP source was written >= 5 instructions earlier, so only the leading s_nop (2 wait states after whatever VALU
    // produced v[]) is needed — the per-step s_nops of a fold-at-a-time formulation cost 24 issue slots per visit.  38 DPP adds.
    float w0 = v[0], w1 = v[1], w2 = v[2], w3 = v[3], w4 = v[4], w5 = v[5], w6 = v[6], w7 = v[7], w8 = v[8], w9 = v[9];
    float w10 = v[10], w11 = v[11], w12 = v[12], w13 = v[13], w14 = v[14], w15 = v[15], w16 = v[16], w17 = v[17], w18 = v[18];      // v[18] = v[19] = 0 (padding): their own fold is skipped
    asm volatile("s_nop 1\n\t"
                 "v_add_f32_dpp %[w0], %[w0], %[w0] row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_f32_dpp %[w2], %[w2], %[w2] row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_f32_dpp %[w4], %[w4], %[w4] row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_f32_dpp %[w6], %[w6], %[w6] row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_f32_dpp %[w8], %[w8], %[w8] row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_f32_dpp %[w10], %[w10], %[w10] row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_f32_dpp %[w12], %[w12], %[w12] row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_f32_dpp %[w14], %[w14], %[w14] row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_f32_dpp %[w16], %[w16], %[w16] row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_f32_dpp %[w0], %[w1], %[w1] row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
                 "v_add_f32_dpp %[w2], %[w3], %[w3] row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
                 "v_add_f32_dpp %[w4], %[w5], %[w5] row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
                 "v_add_f32_dpp %[w6], %[w7], %[w7] row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
                 "v_add_f32_dpp %[w8], %[w9], %[w9] row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
                 "v_add_f32_dpp %[w10], %[w11], %[w11] row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
                 "v_add_f32_dpp %[w12], %[w13], %[w13] row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
                 "v_add_f32_dpp %[w14], %[w15], %[w15] row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
                 "v_add_f32_dpp %[w16], %[w17], %[w17] row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
                 "v_add_f32_dpp %[w0], %[w0], %[w0] row_ror:12 row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_f32_dpp %[w4], %[w4], %[w4] row_ror:12 row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_f32_dpp %[w8], %[w8], %[w8] row_ror:12 row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_f32_dpp %[w12], %[w12], %[w12] row_ror:12 row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_f32_dpp %[w16], %[w16], %[w16] row_ror:12 row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_f32_dpp %[w0], %[w2], %[w2] row_ror:4 row_mask:0xf bank_mask:0xa\n\t"
                 "v_add_f32_dpp %[w4], %[w6], %[w6] row_ror:4 row_mask:0xf bank_mask:0xa\n\t"
                 "v_add_f32_dpp %[w8], %[w10], %[w10] row_ror:4 row_mask:0xf bank_mask:0xa\n\t"
                 "v_add_f32_dpp %[w12], %[w14], %[w14] row_ror:4 row_mask:0xf bank_mask:0xa\n\t"
                 "v_add_f32_dpp %[w16], %[w18], %[w18] row_ror:4 row_mask:0xf bank_mask:0xa\n\t"
                 "v_add_f32_dpp %[w0], %[w0], %[w0] quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_f32_dpp %[w4], %[w4], %[w4] quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_f32_dpp %[w8], %[w8], %[w8] quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_f32_dpp %[w12], %[w12], %[w12] quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_f32_dpp %[w16], %[w16], %[w16] quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_f32_dpp %[w0], %[w0], %[w0] quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_f32_dpp %[w4], %[w4], %[w4] quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_f32_dpp %[w8], %[w8], %[w8] quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_f32_dpp %[w12], %[w12], %[w12] quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_f32_dpp %[w16], %[w16], %[w16] quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf"
                 : [w0] "+v"(w0), [w2] "+v"(w2), [w4] "+v"(w4), [w6] "+v"(w6), [w8] "+v"(w8), [w10] "+v"(w10), [w12] "+v"(w12), [w14] "+v"(w14), [w16] "+v"(w16), [w18] "+v"(w18)
                 : [w1] "v"(w1), [w3] "v"(w3), [w5] "v"(w5), [w7] "v"(w7), [w9] "v"(w9), [w11] "v"(w11), [w13] "v"(w13), [w15] "v"(w15), [w17] "v"(w17));
    z[0] = w0; z[1] = w4; z[2] = w8; z[3] = w12; z[4] = w16;
}
// Wave-wide: the 5 row totals cross rows through permlane swaps (5 swaps + 5 adds): (r0 + r1) + (r2 + r3).
// u0 holds in quad q of row r the wave total of value 4r + {0,2,1,3}[q]; u1 holds in quad q of row 0 value 16 + {0,2,1,3}[q].
__device__ __forceinline__ void wave_reduce20(const float (&v)[NVP], float& u0, float& u1) {
    float z[5];
    row_reduce20(v, z);
    const float t0 = fold16(z[0], z[1]), t1 = fold16(z[2], z[3]), t2 = fold16(z[4], z[4]);
    u0 = fold32(t0, t1);
    u1 = fold32(t2, t2);
}

// Sequential per-pixel state + the 18 per-pair gradient values.  Every upstream gradient enters dL/dalpha only through its
// dot product with this surfel's attributes, so the back-to-front recurrences (colour, depth, alpha, normal, distortion,
// background) collapse into ONE scalar suffix sum
//     X_k = T_final*bg.gC + sum_{i>k} w_i u_i ,  u_i = c_i.gC + d_i g_D + g_A + n_i.gN + dDist/dw_i
//     dL/dalpha_k = T_k u_k - X_k / (1 - alpha_k)
// (algebraically identical to the per-channel "accum_rec" recurrences, 2 state floats instead of 17).
__device__ __forceinline__ void pair_gradients(Pixel& p, const Hit& h, const float4 q3, const float4 q4, bool ok, int pos, float (&gv)[NVP]) {
    constexpr float MC1 = FAR_N / (FAR_N - NEAR_N), MC2 = (FAR_N * NEAR_N) / (FAR_N - NEAR_N);
    // Branch-free: a pair that was not composited runs the same instructions with alpha = 0 and depth = 1, which leaves T and X
    // unchanged (x 1, + 0); its u and dL/dalpha are forced to zero.
    const float alpha = ok ? h.alpha : 0.f, depth = ok ? h.depth : 1.f;
    const float i1a = SURFEL_RCP(1.f - alpha);
    p.T = p.T * i1a;
    const float w = alpha * p.T;
    const float inv_d = SURFEL_RCP(depth);
    const float mm = FMA(-(MC1 * NEAR_N), inv_d, MC1);
    float u = FMA(FMA(mm, FMA(mm, p.final_A, -2.f * p.fM1), p.fM2), p.g_dist, p.g_alpha);
    u = FMA(q3.w, p.gC0, u); u = FMA(q4.x, p.gC1, u); u = FMA(q4.y, p.gC2, u);
    u = FMA(depth, p.g_depth, u);
    u = FMA(q3.x, p.gN0, u); u = FMA(q3.y, p.gN1, u); u = FMA(q3.z, p.gN2, u);
    u = ok ? u : 0.f;          // an idle row may be looking at a stale LDS record: nothing of it may reach the pixel's state
    const float dL_dalpha = ok ? FMA(p.T, u, -(p.X * i1a)) : 0.f;
    p.X = FMA(w, u, p.X);
    float dL_dz = FMA((2.f * w * p.g_dist) * FMA(mm, p.final_A, -p.fM1), MC2 * inv_d * inv_d, w * p.g_depth);
    dL_dz += (ok & (pos == p.medc)) ? p.g_med : 0.f;
    gv[15] = w * p.gC0; gv[16] = w * p.gC1; gv[17] = w * p.gC2;
    gv[11] = w * p.gN0; gv[12] = w * p.gN1; gv[13] = w * p.gN2;
    gv[14] = h.G * dL_dalpha;
    gv[18] = 0.f; gv[19] = 0.f;
    const float nGG = -h.G * (h.opa * dL_dalpha);      // dL/dG * dG/drho*2 ; 0.99 clamp is pass-through
    // low-pass branch: no gradient reaches the intersection.  The selects zero (s, 1/p2) themselves —
    // they may be inf there (p2 ~ 0 on edge-on discs) and 0 * inf must not enter the sums.
    const float sxg = h.use3d ? h.sx : 0.f, syg = h.use3d ? h.sy : 0.f, ipg = h.use3d ? h.ip : 0.f;
    const float g2 = h.use3d ? 0.f : nGG * FILTER_INV_SQUARE;
    const float ax = FMA(nGG, sxg, dL_dz * h.Twx) * ipg, ay = FMA(nGG, syg, dL_dz * h.Twy) * ipg;
    const float dp2 = -FMA(ax, sxg, ay * syg);
    // -dk = dp x l ,  -dl = k x dp
    const float nk0 = FMA(ay, h.lz, -(dp2 * h.ly)), nk1 = FMA(dp2, h.lx, -(ax * h.lz)), nk2 = FMA(ax, h.ly, -(ay * h.lx));
    const float nl0 = FMA(h.ky, dp2, -(h.kz * ay)), nl1 = FMA(h.kz, ax, -(h.kx * dp2)), nl2 = FMA(h.kx, ay, -(h.ky * ax));
    gv[0] = nk0; gv[1] = nk1; gv[2] = nk2;
    gv[3] = nl0; gv[4] = nl1; gv[5] = nl2;
    gv[6] = FMA(dL_dz, sxg, -FMA(p.pxf, nk0, p.pyf * nl0));
    gv[7] = FMA(dL_dz, syg, -FMA(p.pxf, nk1, p.pyf * nl1));
    gv[8] = dL_dz - FMA(p.pxf, nk2, p.pyf * nl2);
    gv[9] = g2 * h.dx; gv[10] = g2 * h.dy;
}

// ---------------------------------------------------------------------------------------------
// blend_bwd, rows variant
// ---------------------------------------------------------------------------------------------
// Round 5 (what the forward got in round 4, profiles/r04_wg_trace.md section 5b):
//   * staging from the forward's TILE STREAM (surfel_common.h) where the frame has one: all 256 threads read the batch's 10 KB of
//     records and its footprint bits as contiguous 16-B pieces — requested before the previous batch's last flush, consumed behind it —
//     instead of ids -> 112-B gather -> footprint test; the gather path stays for frames without a stream (same bits);
//   * the walk: every row expands its 128-bit instance mask ONCE per batch into a byte list (the row's 16 lanes take one mask byte
//     each, DPP prefix sum); a round's visits are a contiguous run of that list, the loop is a counted loop over the wave's longest
//     run (no find-first-set / clear / word refill / votes per visit).
// Per-pair arithmetic, slot assignment and the flush's summation tree are unchanged: rows / quad stay bit-identical.
constexpr int LROW = BS + 4;      // bytes per row list, padded to whole words

template <bool STATS, bool STREAM>
__global__ void __launch_bounds__(BLOCK, ROWS_WAVES) blend_bwd_rows_kernel(BlendBwdArgs a) {
    __shared__ float4 s_rec[BS * 5];                          // 10 KB: q0-q4 of the staged instances
    __shared__ float4 s_slot[NSLOT * 5];                      // 21.25 KB: row totals, one 80-B slot per (instance, sub-tile)
    __shared__ uint32_t s_info[BS];                           // overlap bits (row order) | wave-local exclusive slot prefix << 16
    __shared__ unsigned long long s_rmask[16][BS / 64];       // per row: the staged instances that reach its sub-tile
    __shared__ uint32_t s_wtot[BS / 64];                      // slots of each staging wave's 64 instances
    __shared__ int s_rowlast[16];                             // per row: the largest `last` of its 16 pixels
    __shared__ __attribute__((aligned(4))) uint8_t s_list[16][LROW];      // per row: staged indices of the instances on its list, back to front
    __shared__ int s_max;
    if (a.scan_rule && device_picks_scan(a)) return;
    if (frame_overflowed(a.n_dev, a.n_cap)) return;
    const int tile = block_tile(a.tile_map, a.map_flag, (int)blockIdx.x, a.gx * a.gy);
    if (tile < 0) return;
    const int tx = tile % a.gx, ty = tile / a.gx;
    int lx, ly, sub;
    thread_pixel(threadIdx.x, lx, ly, sub);
    (void)sub;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int srow = threadIdx.x >> 4;                        // this lane's DPP row among the tile's 16 = its sub-tile's bit
    const int i16 = threadIdx.x & 15;
    const uint32_t below = (1u << srow) - 1u;
    const uint2 range = a.ranges[tile];
    Pixel px = load_pixel(a, tx * TILE + lx, ty * TILE + ly);
    {   // a row never visits an instance behind the last contributor of all its pixels
        int m = px.last;
        m = max(m, __shfl_xor(m, 1)); m = max(m, __shfl_xor(m, 2)); m = max(m, __shfl_xor(m, 4)); m = max(m, __shfl_xor(m, 8));
        if (i16 == 0) s_rowlast[srow] = m;
    }
    for (int k = i16; k < LROW / 4; k += 16) reinterpret_cast<uint32_t*>(&s_list[srow][0])[k] = 0u;      // (bytes past a list's end are read as indices: any valid one will do)
    const int maxc = block_max(px.last, &s_max);              // (its barriers also publish s_rowlast)

    // which value of the row total this lane stores (row_reduce20): lane t of quad q holds value 4t + {0,2,1,3}[q] of z[t]
    const int t4 = lane & 3, quad = (lane >> 2) & 3;
    const int pq = ((quad & 1) << 1) | (quad >> 1);
    float* const s_slotf = reinterpret_cast<float*>(s_slot);
    uint8_t* const lrow = &s_list[srow][0];

    // STREAM: the host found the forward's tile stream for this frame (surfel_api.hip: stream_lookup)
    constexpr bool strm_on = STREAM;
    const float4* __restrict__ strm = a.strm_rec;
    const uint32_t* __restrict__ smask = a.strm_mask;
    float pf_touch = 0.f;                     // gather path: landing register of the prefetch touches (never read)
    uint32_t pf_id = 0;                       // gather path: an id of the next batch (waves 2-3); stream path: the id behind pm (large frames)
    float4 pv0 = make_float4(0.f, 0.f, 0.f, 0.f), pv1 = pv0, pv2 = pv0;      // this thread's pieces of the NEXT batch
    uint32_t pm = 0u;                                                        // ... and (threads 0 .. 127) an instance's footprint bits
    auto fetch = [&](int hi_n) {      // the batch of list positions (hi_n - mbn, hi_n]: ascending positions = ascending stream entries
        const int mbn = min(BS, hi_n), np = STRM_Q * mbn;
        const size_t g0 = (size_t)range.x + (size_t)(hi_n - mbn);
        const float4* __restrict__ src = strm + g0 * STRM_Q;
        if ((int)threadIdx.x < np) pv0 = src[threadIdx.x];
        if ((int)threadIdx.x + BLOCK < np) pv1 = src[threadIdx.x + BLOCK];
        if ((int)threadIdx.x + 2 * BLOCK < np) pv2 = src[threadIdx.x + 2 * BLOCK];
        if ((int)threadIdx.x < mbn) {
            pm = smask[g0 + (size_t)(mbn - 1 - (int)threadIdx.x)];
            if (a.has_rec) pf_id = a.point_list[g0 + (size_t)(mbn - 1 - (int)threadIdx.x)];      // (large frames: the surfel that gets a record, for its "has a record" byte)
        }
    };
    if (strm_on && maxc > 0) fetch(maxc);

    for (int hi = maxc; hi > 0; hi -= BS) {
        const int mb = min(BS, hi);
        if (!strm_on) asm volatile("s_waitcnt vmcnt(0)" : "+v"(pf_touch) :: "memory");      // the previous touch has landed: pf_touch may be rewritten
        __syncthreads();                      // previous batch fully flushed
        unsigned ovr = 0;
        if (strm_on) {
            // pieces -> s_rec.  Stream entry jj (ascending position) is staged instance t = mb - 1 - jj (t = 0: the deepest position)
            const int np = STRM_Q * mb;
#pragma unroll
            for (int k = 0; k < 3; k++) {
                const int p = (int)threadIdx.x + BLOCK * k;
                const int jj = (p * 13108) >> 16;       // p / 5 for p < 640
                if (p < np) s_rec[(mb - 1 - jj) * 5 + (p - 5 * jj)] = k == 0 ? pv0 : (k == 1 ? pv1 : pv2);
            }
            if ((int)threadIdx.x < mb) {
                ovr = pm & 0xffffu;
                if (a.has_rec) a.has_rec[pf_id] = 1;      // every staged instance gets a record (finish_tail)
            }
        } else if (wave < BS / 64) {
            if ((int)threadIdx.x < mb) {
                const int pos = hi - (int)threadIdx.x;
                const uint32_t id = a.point_list[range.x + pos - 1];
                const float4* __restrict__ src = reinterpret_cast<const float4*>(a.rec + (size_t)id * REC_F);
                if (a.has_rec) a.has_rec[id] = 1;
                const float4 v0 = src[0], v1 = src[1], v2 = src[2], v3 = src[3], v4 = src[4], v5 = src[5], v6 = src[6];
                s_rec[threadIdx.x * 5 + 0] = v0; s_rec[threadIdx.x * 5 + 1] = v1; s_rec[threadIdx.x * 5 + 2] = v2;
                s_rec[threadIdx.x * 5 + 3] = v3; s_rec[threadIdx.x * 5 + 4] = v4;
                ovr = subtile_overlap_rows(make_foot(v2, v5, v6), tx * TILE, ty * TILE);
            }
        } else if (hi > BS) {
            // gather path, waves 2-3: fetch the NEXT batch's ids, and after the barrier (while everybody walks) pull those records
            // towards this XCD's L2 with one dword load each, so that the next gather overlaps this batch's arithmetic
            const int t = (int)threadIdx.x - BS;
            if (t < min(BS, hi - BS)) pf_id = a.point_list[range.x + (hi - BS - t) - 1];
        }
        if (wave < BS / 64) {
            if ((int)threadIdx.x < mb) {
                const int pos = hi - (int)threadIdx.x;
                unsigned live = 0;
#pragma unroll
                for (int s = 0; s < 16; s++) live |= (pos <= s_rowlast[s]) ? (1u << s) : 0u;
                ovr &= live;
            }
            const int cnt = __popc(ovr);
            int incl = cnt;                   // inclusive prefix of the slot counts over the wave
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { const int v = __shfl_up(incl, o); if (lane >= o) incl += v; }
            s_info[threadIdx.x] = ovr | ((uint32_t)(incl - cnt) << 16);
            if (lane == 63) s_wtot[wave] = (uint32_t)incl;
#pragma unroll
            for (int s = 0; s < 16; s++) {
                const unsigned long long b = __ballot((ovr >> s) & 1u);
                if (lane == 0) s_rmask[s][wave] = b;
            }
        }
        __syncthreads();
        if (!strm_on && wave >= BS / 64 && hi > BS && (int)threadIdx.x - BS < min(BS, hi - BS)) {
            const float* ptr = a.rec + (size_t)pf_id * REC_F;
            asm volatile("global_load_dword %0, %1, off" : "+v"(pf_touch) : "v"(ptr) : "memory");
        }
        const int off1 = (int)s_wtot[0];                      // slots ahead of the second staging wave's instances
        const int tot = off1 + (int)s_wtot[1];
        const int nrounds = max(1, (tot + RSLOTS - 1) / RSLOTS);
        // round of every staged instance (lane -> instances lane and 64 + lane): the one its first slot falls into; an instance
        // without slots belongs to the round of its position (clamped) and only gets its zero record written there
        const int rnd0 = min((int)(s_info[lane] >> 16) >> RSHIFT, nrounds - 1);
        const int rnd1 = min(((int)(s_info[64 + lane] >> 16) + off1) >> RSHIFT, nrounds - 1);
        const unsigned long long c0 = s_rmask[srow][0], c1 = s_rmask[srow][1];
        {   // the row's byte list: lane i16 expands mask byte i16 (instances 8 i16 .. 8 i16 + 7) behind an exclusive prefix over the row
            const uint32_t wsel = (i16 & 8) ? ((i16 & 4) ? (uint32_t)(c1 >> 32) : (uint32_t)c1) : ((i16 & 4) ? (uint32_t)(c0 >> 32) : (uint32_t)c0);
            const uint32_t byte = (wsel >> (8 * (i16 & 3))) & 0xffu;
            int off = __popc(byte), t;
            t = __builtin_amdgcn_update_dpp(0, off, 0x111, 0xf, 0xf, true); off += t;      // row_shr:1
            t = __builtin_amdgcn_update_dpp(0, off, 0x112, 0xf, 0xf, true); off += t;      // row_shr:2
            t = __builtin_amdgcn_update_dpp(0, off, 0x114, 0xf, 0xf, true); off += t;      // row_shr:4
            t = __builtin_amdgcn_update_dpp(0, off, 0x118, 0xf, 0xf, true); off += t;      // row_shr:8
            off -= __popc(byte);
#pragma unroll
            for (int k = 0; k < 8; k++)
                if ((byte >> k) & 1u) lrow[off + __popc(byte & ((1u << k) - 1u))] = (uint8_t)(8 * i16 + k);
        }
        int lpos = 0;                                         // list entries of this row consumed by the rounds so far
        unsigned long long cum0 = 0ull, cum1 = 0ull;          // instances of rounds <= r
        for (int r = 0; r < nrounds; r++) {
            const unsigned long long m0 = __ballot((lane < mb) & (rnd0 == r)), m1 = __ballot((64 + lane < mb) & (rnd1 == r));
            cum0 |= m0; cum1 |= m1;
            // ---- walk: every row visits, in list order, its instances of round r — a contiguous run of its byte list (an instance's
            // round grows with its staged index)
            const int lend = __popcll(c0 & cum0) + __popcll(c1 & cum1);
            const int len = lend - lpos;
            const int niter = max(max(__builtin_amdgcn_readlane(len, 0), __builtin_amdgcn_readlane(len, 16)),
                                  max(__builtin_amdgcn_readlane(len, 32), __builtin_amdgcn_readlane(len, 48)));
            if (niter > 0) {
                // (No read-ahead of the next visit's record: measured in round 5 with the forward's scheme — next record in registers, loop
                // unrolled by two with the roles swapped — it buys nothing where this walk is the default (C2: 0.1988 vs 0.1991 ms, four
                // waves per SIMD cover the LDS latency) and 1 - 2 % on the frames the scan walk takes anyway, for 21 registers.)
                for (int v = 0; v < niter; v++) {
                    const bool actc = v < len;
                    const int jc = (int)lrow[actc ? lpos + v : 0];      // (past a shorter row's run the visit is masked: it reads entry 0, inside the list)
                    const uint32_t infoc = s_info[jc];
                    const float4 c_q0 = s_rec[jc * 5 + 0], c_q1 = s_rec[jc * 5 + 1], c_q2 = s_rec[jc * 5 + 2], c_q3 = s_rec[jc * 5 + 3], c_q4 = s_rec[jc * 5 + 4];
                    Hit h;
                    const int pos = hi - jc;      // 1-based position in the tile's list
                    const bool ok = pair_hit(px, c_q0, c_q1, c_q2, pos, h) & actc;
                    float gv[NVP], z[5];
                    pair_gradients(px, h, c_q3, c_q4, ok, pos, gv);
                    row_reduce20(gv, z);
                    if (STATS) {
                        const unsigned long long okb = __ballot(ok), ab = __ballot(actc);
                        if (lane == 0) {
                            atomicAdd(&a.stats[0], 64ull); atomicAdd(&a.stats[1], (unsigned long long)__popcll(okb));
                            atomicAdd(&a.stats[2], 1ull); atomicAdd(&a.stats[3], (unsigned long long)(__popcll(ab) >> 4));
                            const int hitrows = ((okb & 0xffffull) != 0) + (((okb >> 16) & 0xffffull) != 0) + (((okb >> 32) & 0xffffull) != 0) + ((okb >> 48) != 0);
                            atomicAdd(&a.stats[4], (unsigned long long)hitrows);
                        }
                    }
                    if (actc) {
                        const int E = (int)(infoc >> 16) + (jc >= 64 ? off1 : 0);
                        const int slot = (E & (RSLOTS - 1)) + __popc(infoc & below);
                        float* sp = s_slotf + slot * NVP;
                        const float val = t4 == 0 ? z[0] : (t4 == 1 ? z[1] : (t4 == 2 ? z[2] : z[3]));
                        sp[4 * t4 + pq] = val;
                        if (t4 == 0) sp[16 + pq] = z[4];
                    }
                }
            }
            lpos = lend;
            __syncthreads();
            // the next batch's stream pieces: requested here, in front of the batch's last flush, consumed behind it
            if (strm_on && r == nrounds - 1 && hi > BS) fetch(hi - BS);
            else {      // (no request: tell the compiler the prefetch registers hold nothing from here on — otherwise they stay allocated through every walk)
                asm volatile("" : "=v"(pv0.x), "=v"(pv0.y), "=v"(pv0.z), "=v"(pv0.w), "=v"(pv1.x), "=v"(pv1.y), "=v"(pv1.z), "=v"(pv1.w));
                asm volatile("" : "=v"(pv2.x), "=v"(pv2.y), "=v"(pv2.z), "=v"(pv2.w), "=v"(pm), "=v"(pf_id));
            }
            // ---- flush round r: its instances are a contiguous run of the staged list.  One thread per (instance, float4 of its
            // record): the instance's slots are added in the fixed order ((r0+r1)+(r2+r3)) per wave, waves 0..3 — the summation
            // tree of the quad variant, so both variants give the same bits.
            {
                const int n = __popcll(m0) + __popcll(m1);
                const int t0 = m0 ? __builtin_ctzll(m0) : 64 + (m1 ? __builtin_ctzll(m1) : 0);
                for (int item = threadIdx.x; item < 5 * n; item += BLOCK) {
                    const int k = (item * 13108) >> 16;       // item / 5 for item < 640
                    const int q = item - 5 * k;
                    const int t = t0 + k;
                    const uint32_t inf = s_info[t];
                    const int E = (int)(inf >> 16) + (t >= 64 ? off1 : 0);
                    const float4* sp = s_slot + (E & (RSLOTS - 1)) * 5 + q;
                    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                    for (int w = 0; w < 4; w++) {
                        const unsigned m = (inf >> (4 * w)) & 15u;
                        if (m) {
                            const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
                            float4 v0 = zero, v1 = zero, v2 = zero, v3 = zero;
                            if (m & 1u) { v0 = *sp; sp += 5; }
                            if (m & 2u) { v1 = *sp; sp += 5; }
                            if (m & 4u) { v2 = *sp; sp += 5; }
                            if (m & 8u) { v3 = *sp; sp += 5; }
                            acc = add4(acc, add4(add4(v0, v1), add4(v2, v3)));
                        }
                    }
                    const float4 r4 = s_rec[t * 5 + 4];
                    reinterpret_cast<float4*>(a.grec + grec_slot(r4, tx, ty) * GREC_F)[q] = acc;
                }
            }
            __syncthreads();                  // slots reusable
        }
    }
    finish_tail(a, range, maxc, tile, tx, ty);
}

// ---------------------------------------------------------------------------------------------
// blend_bwd, quad variant (round 1's kernel on the shared per-pair arithmetic)
// ---------------------------------------------------------------------------------------------
template <bool STATS>
__global__ void __launch_bounds__(BLOCK) blend_bwd_quad_kernel(BlendBwdArgs a) {
    __shared__ float4 s_rec[BS * 5];                 // 10 KB
    __shared__ float s_acc[4][BB][NVP];              // 20 KB: per-wave partial sums of the current sub-batch
    __shared__ unsigned long long s_mask[4];         // which sub-batch slots each wave wrote
    __shared__ unsigned long long s_qmask[4][4];     // [quad][staging wave] overlap bitmasks of the staged batch
    __shared__ int s_quadlast[4];                    // per quad (= wave): the largest `last` of its 64 pixels
    __shared__ int s_max;
    if (a.scan_rule && device_picks_scan(a)) return;
    if (frame_overflowed(a.n_dev, a.n_cap)) return;
    const int tile = block_tile(a.tile_map, a.map_flag, blockIdx.x, a.gx * a.gy);
    if (tile < 0) return;
    const int tx = tile % a.gx, ty = tile / a.gx;
    int lx, ly, sub;
    thread_pixel(threadIdx.x, lx, ly, sub);
    (void)sub;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint2 range = a.ranges[tile];
    Pixel px = load_pixel(a, tx * TILE + lx, ty * TILE + ly);
    {   // a quad never visits an instance behind the last contributor of all its pixels (those visits would find no lane to work on)
        int m = px.last;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m = max(m, __shfl_xor(m, o));
        if (lane == 0) s_quadlast[wave] = m;
    }
    const int maxc = block_max(px.last, &s_max);              // (its barriers also publish s_quadlast)

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
                if (a.has_rec) a.has_rec[id] = 1;
                const float4 v0 = src[0], v1 = src[1], v2 = src[2], v3 = src[3], v4 = src[4], v5 = src[5], v6 = src[6];
                s_rec[threadIdx.x * 5 + 0] = v0; s_rec[threadIdx.x * 5 + 1] = v1; s_rec[threadIdx.x * 5 + 2] = v2;
                s_rec[threadIdx.x * 5 + 3] = v3; s_rec[threadIdx.x * 5 + 4] = v4;
                ov = quad_overlap(make_foot(v2, v5, v6), tx * TILE, ty * TILE);
                const int pos = hi - (int)threadIdx.x;
#pragma unroll
                for (int qd = 0; qd < 4; qd++) ov &= (pos <= s_quadlast[qd]) ? ~0u : ~(1u << qd);
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
                Hit h;
                const bool ok = pair_hit(px, q0, q1, q2, pos, h);
                const unsigned long long okb = __ballot(ok);
                if (STATS) {
                    if (lane == 0) {
                        atomicAdd(&a.stats[0], 64ull); atomicAdd(&a.stats[1], (unsigned long long)__popcll(okb));
                        atomicAdd(&a.stats[2], 1ull); atomicAdd(&a.stats[3], 1ull);
                        atomicAdd(&a.stats[4], (unsigned long long)(okb != 0ull));
                        const int hitrows = ((okb & 0xffffull) != 0) + (((okb >> 16) & 0xffffull) != 0) + (((okb >> 32) & 0xffffull) != 0) + ((okb >> 48) != 0);
                        atomicAdd(&a.stats[5], (unsigned long long)hitrows);
                    }
                }
                if (okb == 0ull) continue;      // wave-uniform
                const float4 q3 = s_rec[j * 5 + 3], q4 = s_rec[j * 5 + 4];
                float gv[NVP];
                pair_gradients(px, h, q3, q4, ok, pos, gv);
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
                float4 out[5];
#pragma unroll
                for (int q = 0; q < 5; q++) out[q] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int w = 0; w < 4; w++) {
                    if (s_mask[w] & bit) {
#pragma unroll
                        for (int q = 0; q < 5; q++) out[q] = add4(out[q], *reinterpret_cast<const float4*>(&s_acc[w][jj][4 * q]));
                    }
                }
                const float4 q4 = s_rec[(sb * BB + jj) * 5 + 4];
                float4* __restrict__ dst = reinterpret_cast<float4*>(a.grec + grec_slot(q4, tx, ty) * GREC_F);
#pragma unroll
                for (int q = 0; q < 5; q++) dst[q] = out[q];
            }
            __syncthreads();                  // s_acc / s_mask reusable
        }
    }
    finish_tail(a, range, maxc, tile, tx, ty);
}

void launch_blend_bwd(const BlendBwdArgs& a, hipStream_t s) {
    const dim3 grid(a.map_len), block(BLOCK);
    if (a.variant == 3) { launch_blend_bwd_scan(a, s); return; }
    if (a.scan_rule) launch_blend_bwd_scan(a, s);      // (returns at once unless the device rule picks it; the kernel below does the opposite)
    if (a.variant == 0) {
        const bool st = a.strm_rec != nullptr;
        if (a.stats) { if (st) hipLaunchKernelGGL((blend_bwd_rows_kernel<true, true>), grid, block, 0, s, a); else hipLaunchKernelGGL((blend_bwd_rows_kernel<true, false>), grid, block, 0, s, a); }
        else { if (st) hipLaunchKernelGGL((blend_bwd_rows_kernel<false, true>), grid, block, 0, s, a); else hipLaunchKernelGGL((blend_bwd_rows_kernel<false, false>), grid, block, 0, s, a); }
    } else {
        if (a.stats) hipLaunchKernelGGL(blend_bwd_quad_kernel<true>, grid, block, 0, s, a);
        else hipLaunchKernelGGL(blend_bwd_quad_kernel<false>, grid, block, 0, s, a);
    }
}

}  // namespace surfel
