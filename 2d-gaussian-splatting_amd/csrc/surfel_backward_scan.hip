// surfel_backward_scan.hip — blend_bwd, "scan" walk: LANES ARE INSTANCES, pixels are the steps (gfx950).
//
// The rows / quad walks (surfel_backward.hip) keep a pixel per lane and pay a 38-DPP cross-lane reduction plus an LDS slot per
// (instance, sub-tile) visit to turn 16 per-pixel contributions into one per-instance total — 20-25 % of a visit's issue slots,
// on kernels that are bound by VALU issue.  Here the roles are swapped:
//   * every DPP row of 16 lanes still owns one 4x4-pixel sub-tile and ITS OWN list of the staged instances whose alpha >= 1/255
//     footprint reaches it — but a lane now holds one INSTANCE of that list (16 consecutive ones = a chunk), with the instance's
//     record and its 18 gradient accumulators in registers, and the row steps through the sub-tile's 16 pixels;
//   * the per-pixel back-to-front recurrences (T_k = T_{k+1} / (1 - alpha_k), X_k = X_{k+1} + w_k u_k; see pair_gradients in
//     surfel_backward.hip for the two-scalar formulation) become two 16-lane DPP SCANS per step (4 v_mul_f32_dpp + 5 v_add_f32_dpp),
//     the pixel's state (T, X) lives in LDS next to its upstream gradients and is advanced by the row's last lane;
//   * gradients accumulate in the owning lane's registers as fused multiply-adds — no reduction tree, no per-visit slot.
// After a chunk (16 pixel steps) a lane parks its partial record in an LDS slot; the slots of a round (chunk c of all 16 lists)
// are added into the instance's totals by a fixed thread in a fixed order ((round, sub-tile) ascending), and every staged instance's
// 80-B gradient record is written once per batch — no atomics, bit-reproducible run to run.  The summation order differs from the
// rows / quad walks, so this walk is NOT bit-identical to them; it is held to the same oracle bars instead
// (tests/test_gpu_parity.py::test_scan_walk_*).
// Semantics: oracle/surfel_oracle.c stages 4-5 (restating the absent diff-surfel-rasterization).
#include "surfel_blend_bwd.h"

namespace surfel {

namespace {

constexpr int SB = 128;      // instances staged per batch (one 112-B gather per thread of waves 0-1)
constexpr int CH = 16;       // instances per chunk = lanes of a DPP row

// inclusive product / exclusive sum over the 16 lanes of every DPP row, lane 0 first
__device__ __forceinline__ float row_scan_mul(float x) {
    // v_mul_f32_dpp without bound_ctrl: lanes whose source lies outside the row keep their value.  2 wait states between the
    // VALU write of a register and its use as a DPP source (the compiler cannot see into the asm).
    asm(
                 "s_nop 2\n\t"
                 "v_mul_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\t"
                 "v_mul_f32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\t"
                 "v_mul_f32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\t"
                 "v_mul_f32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf"
                 : "+v"(x));
    return x;
}
template <int N>
__device__ __forceinline__ float row_shr_add(float x) {      // x + (x shifted right by N lanes inside each row, zero fill): one v_add_f32_dpp
    return x + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x110 + N, 0xf, 0xf, true));
}
__device__ __forceinline__ float row_scan_add_incl(float x) {
    x = row_shr_add<1>(x); x = row_shr_add<2>(x); x = row_shr_add<4>(x); x = row_shr_add<8>(x);
    return x;
}
__device__ __forceinline__ float row_scan_add_excl(float x) {
    float e = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x111, 0xf, 0xf, true));      // row_shr:1, zero fill
    e = row_shr_add<1>(e); e = row_shr_add<2>(e); e = row_shr_add<4>(e); e = row_shr_add<8>(e);
    return e;
}

// index (0 .. 127) of set bit number k (0-based, k < popcount) of the 128-bit word (w1 : w0)
__device__ __forceinline__ int select_bit128(unsigned long long w0, unsigned long long w1, int k) {
    const int c0 = __popcll(w0);
    unsigned long long w = k < c0 ? w0 : w1;
    int idx = k < c0 ? 0 : 64;
    k = k < c0 ? k : k - c0;
#pragma unroll
    for (int sh = 32; sh > 0; sh >>= 1) {
        const unsigned long long lo = w & ((1ull << sh) - 1ull);
        const int c = __popcll(lo);
        if (k >= c) { k -= c; w >>= sh; idx += sh; } else w = lo;
    }
    return idx;
}

}  // namespace

// [r6] BATCH TRUNCATION.  A batch lasts (chunks of its longest sub-tile list) rounds, and a round lasts one 16-step walk whatever it
// holds — a workgroup's round is as long as ONE walk however many of its lanes and waves take part (profiles/r06_negative_results.md
// section 6): a longest list of 50 costs four rounds, the fourth for two instances.  Where that pays (rounds + half a round of fixed
// cost per staged instance gets smaller) the batch is cut in front of the staged instance that opens the longest list's last, partial
// chunk (the deepest `keep` instances are walked: full chunks on the longest list) and the next batch starts with the rest — staged
// again from there, nothing is carried.
// Where a batch ends follows from the frame's lists alone, the same with and without the tile stream: same bits either way.
// Measured (profiles/r06_ab_scan_trunc.jsonl): blend_bwd garden 1.345 -> 1.296 ms, C4 0.977 -> 0.944 with a fixed "1 .. 8 instances into
// the chunk" rule; the cost rule above another 0.6 - 0.9 % (r06_ab_scan_trunc_policy.jsonl).
// (The two staging paths are kept apart with `if constexpr`: an earlier form that shared a lambda between them cost the STREAM
// instantiation 8 % — same instruction counts, another schedule.)
#ifndef SCAN_TRUNC_REM
#define SCAN_TRUNC_REM 1      // 0: fixed 128-position batches
#endif

// LDS records of a pixel, indexed by the pixel's owner thread (tid = 16 * sub-tile row + pixel of the sub-tile):
//   s_pix  (read-only in the walk, 3 x float4): [0] gC0 gC1 gC2 g_depth   [1] gN0 gN1 gN2 g_med   [2] a2 a1 a0 last
//   s_state (float4): T X medc -      (T, X) advanced once per chunk by the row's last lane
// with the distortion / alpha terms of u pre-multiplied:  u = mm (mm a2 + a1) + a0 + c.gC + depth g_depth + n.gN,
//   a2 = final_A g_dist,  a1 = -2 M1 g_dist,  a0 = M2 g_dist + g_alpha.
// The walk of a chunk is ONE basic block: any branch inside it (a predicated state write, a conditional slot write) lets LLVM sink
// the gradient half of all sixteen steps behind the last step and spill their intermediates.  So the state write is issued by
// every lane — the last lane of a row writes the pixel's state, the others a scratch area behind it — and idle lanes park a
// (never read) slot as well.
constexpr int STATE_SCRATCH = 49;      // float4s: 16 steps x 16 B + 64 lanes x 8 B
// The 4 rows of a wave read 4 different pixels' records in one instruction (a broadcast inside each row).  With the natural row
// strides (16 x 48 B = 192 dwords, 16 x 16 B = 64 dwords: both 0 mod 64 banks) the four addresses fell on the same banks — a 4-way
// conflict on every read of every step: 16.2 M of the kernel's LDS conflict cycles per launch at C2 against 1.4 M for the rows walk
// (profiles/r03gscan_C2_pmc.json).  One float4 of padding per row moves the rows 4 banks apart.
constexpr int PIX_ROW = 16 * 3 + 1, STATE_ROW = 16 + 1;
#ifndef SCAN_INCL
#define SCAN_INCL 1      // 1: inclusive sum scan (4 DPP adds) and one subtraction instead of the exclusive one (mov + 4 adds)
#endif
#ifndef SCAN_AFFINE
#define SCAN_AFFINE 1    // 1: the plane gradients are summed as three moments of dL/dp and turned into dL/dTu, dL/dTv, dL/dTw once per chunk
#endif
#ifndef SCAN_MIN_WG
#define SCAN_MIN_WG 3
#endif
template <bool STATS, bool STREAM>
__global__ void __launch_bounds__(BLOCK, SCAN_MIN_WG) blend_bwd_scan_kernel(BlendBwdArgs a) {
    __shared__ float4 s_rec[SB * 5];                          // 10 KB: q0-q4 of the staged instances
    __shared__ float4 s_slot[BLOCK * 5];                      // 20 KB: the round's partial records, one per walking lane
    __shared__ float4 s_pix[16 * PIX_ROW];                    // 12.3 KB
    __shared__ float4 s_state[16 * STATE_ROW + STATE_SCRATCH];      // 5.1 KB
    __shared__ unsigned long long s_bal[16][SB / 64];         // per sub-tile: the staged instances on its list
    __shared__ uint8_t s_list[16][SB];                        // per sub-tile: its list (staged indices, back to front)
    __shared__ uint4 s_rank[SB];                              // per staged instance: its rank on each of the 16 lists (0xff: not on it)
    __shared__ uint32_t s_cmm[SB];                            // per staged instance: first | last << 8 round it takes part in
    __shared__ int s_rowlast[16];
    __shared__ int s_max;
    const int tid = threadIdx.x;
    if (a.scan_rule && a.variant != 3 && !device_picks_scan(a)) return;
    if (frame_overflowed(a.n_dev, a.n_cap)) return;
    const int tile = block_tile(a.tile_map, a.map_flag, blockIdx.x, a.gx * a.gy);
    if (tile < 0) return;
    const int tx = tile % a.gx, ty = tile / a.gx;
    int lx, ly, sub;
    thread_pixel(tid, lx, ly, sub);
    (void)sub;
    const int wave = tid >> 6, lane = tid & 63, i16 = tid & 15;
    const int srow = tid >> 4;                                // this lane's DPP row among the tile's 16 = its sub-tile's bit
    const uint2 range = a.ranges[tile];
    {
        const Pixel px = load_pixel(a, tx * TILE + lx, ty * TILE + ly);
        int m = px.last;
        m = max(m, __shfl_xor(m, 1)); m = max(m, __shfl_xor(m, 2)); m = max(m, __shfl_xor(m, 4)); m = max(m, __shfl_xor(m, 8));
        if (i16 == 0) s_rowlast[srow] = m;
        float4* const mine = s_pix + srow * PIX_ROW + i16 * 3;
        mine[0] = make_float4(px.gC0, px.gC1, px.gC2, px.g_depth);
        mine[1] = make_float4(px.gN0, px.gN1, px.gN2, px.g_med);
        mine[2] = make_float4(px.final_A * px.g_dist, -2.f * px.fM1 * px.g_dist, FMA(px.fM2, px.g_dist, px.g_alpha), __int_as_float(px.last));
        s_state[srow * STATE_ROW + i16] = make_float4(px.T, px.X, __int_as_float(px.medc), 0.f);
        const int maxc0 = block_max(px.last, &s_max);         // (its barriers also publish s_rowlast and s_pix)
        (void)maxc0;
    }
    const int maxc = s_max;
    const float sx0 = (float)(tx * TILE + (lx & ~3)), sy0 = (float)(ty * TILE + (ly & ~3));      // the sub-tile's first pixel
    const float4* const prow = s_pix + srow * PIX_ROW;        // the sub-tile's 16 pixel records
    const float4* const srd = s_state + srow * STATE_ROW;     // ... and states
    // where this lane's state writes go: the row's last lane advances the pixel's (T, X), every other lane hits scratch
    float2* const swr = i16 == 15 ? reinterpret_cast<float2*>(s_state + srow * STATE_ROW) : reinterpret_cast<float2*>(s_state + 16 * STATE_ROW) + lane;

    // flush ownership: thread (t, half) adds up values [0, 12) or [12, 20) of staged instance t
    const int ft = tid & (SB - 1), fh = tid >> 7;
    float4 f0 = make_float4(0.f, 0.f, 0.f, 0.f), f1 = f0, f2 = f0;

    // Software-pipelined staging: what batch b+1 needs is loaded into registers while batch b is walked; the gradient records of
    // batch b are stored at the top of batch b+1, BEHIND the consumption of the prefetched registers (gfx9 counts loads and stores in
    // one vmcnt: a wait for the loads would otherwise wait for the stores as well).
    //   * frames with a TILE STREAM (surfel_common.h; round 5): every thread holds up to three contiguous 16-B pieces of the batch's
    //     10 KB of stream records, the threads of the upper half also an instance's footprint bits — no ids, no gather, no footprint test;
    //   * frames without one: half a 112-B record per thread ((instance ft, half fh) — fh 0: q0 q1 q3 q4, fh 1: q2 q5 q6 and the
    //     footprint test), the surfel ids one batch further ahead.
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 pa = zero4, pb = zero4, pc = zero4, pd = zero4;
    uint32_t nid = 0;
    constexpr bool TRUNC = SCAN_TRUNC_REM > 0;
    constexpr bool strm_on = STREAM;      // the host found the forward's tile stream for this frame (surfel_api.hip: stream_lookup)
    const float4* __restrict__ strm = a.strm_rec;
    const uint32_t* __restrict__ smask = a.strm_mask;
    auto fetch = [&](int hi_n) {      // the batch of list positions (hi_n - mbn, hi_n]: pa pb pc = pieces tid, tid + 256, tid + 512; nid = footprint bits
        const int mbn = min(SB, hi_n), np = STRM_Q * mbn;
        const size_t g0 = (size_t)range.x + (size_t)(hi_n - mbn);
        const float4* __restrict__ src = strm + g0 * STRM_Q;
        if (tid < np) pa = src[tid];
        if (tid + BLOCK < np) pb = src[tid + BLOCK];
        if (tid + 2 * BLOCK < np) pc = src[tid + 2 * BLOCK];
        if (ft < mbn) {      // upper half: the instance's footprint bits; lower half (large frames): its surfel, for the "has a record" byte
            if (fh == 1) nid = smask[g0 + (size_t)(mbn - 1 - ft)];
            else if (a.has_rec) nid = a.point_list[g0 + (size_t)(mbn - 1 - ft)];
        }
    };
    if (strm_on) {
        if (maxc > 0) fetch(maxc);
    } else {
        const float4* __restrict__ recq = reinterpret_cast<const float4*>(a.rec);
        if (ft < min(SB, maxc)) {
            const uint32_t id = a.point_list[range.x + (maxc - ft) - 1];
            const float4* __restrict__ src = recq + (size_t)id * REC_Q;
            if (a.has_rec && fh == 0) a.has_rec[id] = 1;      // every staged instance gets a record (finish_tail)
            if (fh == 0) { pa = src[0]; pb = src[1]; pc = src[3]; pd = src[4]; } else { pa = src[2]; pb = src[5]; pc = src[6]; }
        }
        if (!TRUNC && maxc - SB > 0 && ft < min(SB, maxc - SB)) nid = a.point_list[range.x + (maxc - SB - ft) - 1];
    }
    bool pend = false;
    size_t pend_slot = 0;
    int step = SB;
    for (int hi = maxc; hi > 0; hi -= step) {
        const int mb = min(SB, hi);
        __syncthreads();                      // previous batch written out: s_rec / s_list / s_rank reusable
        unsigned ovr = 0;
        if (strm_on) {
            // pieces -> s_rec: stream entry jj (ascending position) is staged instance mb - 1 - jj (0 = the deepest position)
            const int np = STRM_Q * mb;
#pragma unroll
            for (int k = 0; k < 3; k++) {
                const int p = tid + BLOCK * k;
                const int jj = (p * 13108) >> 16;       // p / 5 for p < 640
                if (p < np) s_rec[(mb - 1 - jj) * 5 + (p - 5 * jj)] = k == 0 ? pa : (k == 1 ? pb : pc);
            }
            if (ft < mb) {
                if (fh == 1) {
                    ovr = nid & 0xffffu;
                    const int pos = hi - ft;
                    unsigned live = 0;            // a sub-tile never meets an instance behind the last contributor of all its pixels
#pragma unroll
                    for (int s = 0; s < 16; s++) live |= (pos <= s_rowlast[s]) ? (1u << s) : 0u;
                    ovr &= live;
                } else if (a.has_rec) a.has_rec[nid] = 1;      // every staged instance gets a record (finish_tail)
            }
        } else if (ft < mb) {
            if (fh == 0) { s_rec[ft * 5 + 0] = pa; s_rec[ft * 5 + 1] = pb; s_rec[ft * 5 + 3] = pc; s_rec[ft * 5 + 4] = pd; }
            else {
                s_rec[ft * 5 + 2] = pa;
                ovr = subtile_overlap_rows(make_foot(pa, pb, pc), tx * TILE, ty * TILE);
                const int pos = hi - ft;
                unsigned live = 0;            // a sub-tile never meets an instance behind the last contributor of all its pixels
#pragma unroll
                for (int s = 0; s < 16; s++) live |= (pos <= s_rowlast[s]) ? (1u << s) : 0u;
                ovr &= live;
            }
        }
        if (pend) {                           // the previous batch's gradient records
            float4* __restrict__ dst = reinterpret_cast<float4*>(a.grec + pend_slot * GREC_F);
            if (fh == 0) { dst[0] = f0; dst[1] = f1; dst[2] = f2; }
            else { dst[3] = f0; dst[4] = f1; }
        }
        f0 = zero4; f1 = zero4; f2 = zero4;
        if (strm_on) {
            if (!TRUNC && hi - SB > 0) fetch(hi - SB);      // (TRUNC: behind the ballots, once this batch's length is known)
        } else if (!TRUNC && hi - SB > 0) {   // next batch's records, and the ids of the one behind it
            const float4* __restrict__ recq = reinterpret_cast<const float4*>(a.rec);
            if (ft < min(SB, hi - SB)) {
                const float4* __restrict__ src = recq + (size_t)nid * REC_Q;
                if (a.has_rec && fh == 0) a.has_rec[nid] = 1;
                if (fh == 0) { pa = src[0]; pb = src[1]; pc = src[3]; pd = src[4]; } else { pa = src[2]; pb = src[5]; pc = src[6]; }
            }
            if (hi - 2 * SB > 0 && ft < min(SB, hi - 2 * SB)) nid = a.point_list[range.x + (hi - 2 * SB - ft) - 1];
        }
        if (fh == 1) {
#pragma unroll
            for (int s = 0; s < 16; s++) {
                const unsigned long long b = __ballot((ovr >> s) & 1u);
                if (lane == 0) s_bal[s][wave - 2] = b;
            }
        }
        __syncthreads();
        int keep = mb;                        // staged instances walked in this batch: the deepest `keep`
        if (TRUNC) {
            const unsigned long long w0 = s_bal[i16][0], w1 = s_bal[i16][1];
            const int n = __popcll(w0) + __popcll(w1);
            int nm = n;
            nm = max(nm, __shfl_xor(nm, 1)); nm = max(nm, __shfl_xor(nm, 2)); nm = max(nm, __shfl_xor(nm, 4)); nm = max(nm, __shfl_xor(nm, 8));
            const int full = nm & ~(CH - 1), rem = nm & (CH - 1);
            if (full > 0 && rem > 0) {
                int ms = n > full ? select_bit128(w0, w1, full) : SB;      // the staged instance that opens this list's chunk behind the full ones
                ms = min(ms, __shfl_xor(ms, 1)); ms = min(ms, __shfl_xor(ms, 2)); ms = min(ms, __shfl_xor(ms, 4)); ms = min(ms, __shfl_xor(ms, 8));
                // cut iff the batch then costs less per staged instance, a batch's fixed part taken as half a round:
                // (r + 1/2) / ms < (r + 3/2) / mb   (r = full chunks of the longest list)
                const int r = full / CH;
                if (ms * (2 * r + 3) > mb * (2 * r + 1)) keep = ms;
            }
            keep = __builtin_amdgcn_readfirstlane(keep);
            if constexpr (STREAM) {
                if (hi - keep > 0) fetch(hi - keep);
            } else {
                // ids, then the half records of the next batch: two dependent round trips under this batch's walk (the ids can no longer
                // be requested a batch ahead: where the next batch starts is only known here)
                if (hi - keep > 0 && ft < min(SB, hi - keep)) {
                    const float4* __restrict__ recq = reinterpret_cast<const float4*>(a.rec);
                    const uint32_t id = a.point_list[range.x + (hi - keep - ft) - 1];
                    const float4* __restrict__ src = recq + (size_t)id * REC_Q;
                    if (a.has_rec && fh == 0) a.has_rec[id] = 1;
                    if (fh == 0) { pa = src[0]; pb = src[1]; pc = src[3]; pd = src[4]; } else { pa = src[2]; pb = src[5]; pc = src[6]; }
                }
            }
            if (ft >= keep) ovr = 0;
        }
        const unsigned long long km0 = keep >= 64 ? ~0ull : ((1ull << keep) - 1ull);
        const unsigned long long km1 = keep >= 128 ? ~0ull : (keep > 64 ? ((1ull << (keep - 64)) - 1ull) : 0ull);
        if (fh == 1) {                        // (the threads that hold the footprints)
            // rank of this instance on every list it is on = instances ahead of it (staged order = back to front) on that list
            uint32_t rk[4] = {~0u, ~0u, ~0u, ~0u};
            uint32_t cmin = 255u, cmax = 0u;
            const unsigned long long lt = (1ull << (ft & 63)) - 1ull;
#pragma unroll
            for (int s = 0; s < 16; s++) {
                const unsigned long long m0 = s_bal[s][0] & km0, m1 = s_bal[s][1] & km1;
                const uint32_t r = ft < 64 ? (uint32_t)__popcll(m0 & lt) : (uint32_t)(__popcll(m0) + __popcll(m1 & lt));
                if ((ovr >> s) & 1u) {
                    s_list[s][r] = (uint8_t)ft;
                    rk[s >> 2] = (rk[s >> 2] & ~(0xffu << (8 * (s & 3)))) | (r << (8 * (s & 3)));
                    cmin = min(cmin, r >> 4); cmax = max(cmax, r >> 4);
                }
            }
            s_rank[ft] = make_uint4(rk[0], rk[1], rk[2], rk[3]);
            s_cmm[ft] = cmin | (cmax << 8);      // an instance on no list: 255 | 0 -> takes part in no round
        }
        // list length of this row's sub-tile, and the number of rounds = chunks of the longest list (the same in every thread)
        const int n_row = __popcll(s_bal[srow][0] & km0) + __popcll(s_bal[srow][1] & km1);
        int nmax = __popcll(s_bal[i16][0] & km0) + __popcll(s_bal[i16][1] & km1);
        nmax = max(nmax, __shfl_xor(nmax, 1)); nmax = max(nmax, __shfl_xor(nmax, 2)); nmax = max(nmax, __shfl_xor(nmax, 4)); nmax = max(nmax, __shfl_xor(nmax, 8));
        const int nrounds = (nmax + CH - 1) / CH;
        __syncthreads();
        const uint32_t fcm = s_cmm[ft];
        const int fcmin = (int)(fcm & 255u), fcmax = (int)(fcm >> 8);

        for (int c = 0; c < nrounds; c++) {
            const int idx = c * CH + i16;
            const bool valid = idx < n_row;
            if (__any(valid)) {
                // ---- chunk c of this row's list: lane = instance
                const int t = valid ? (int)s_list[srow][idx] : 0;
                const float4 q0 = s_rec[t * 5 + 0], q1 = s_rec[t * 5 + 1], q2 = s_rec[t * 5 + 2], q3 = s_rec[t * 5 + 3], q4 = s_rec[t * 5 + 4];
                const int pos = valid ? hi - t : 0x7fffffff;      // 1-based position in the tile's list; an idle lane composites nothing
                const float Twx = q1.z, Twy = q1.w, Twz = q2.x, opa = q2.w;
                // planes of the sub-tile's 4 pixel columns (the forward's k = px Tw - Tu, same rounding); the rows' planes
                // (l = py Tw - Tv) are set up once per pixel row below — 16 registers instead of 40 for all eight
                float K[4][3], DX[4];
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    Hit hh;
                    pair_planes(sx0 + (float)e, sy0, q0, q1, q2, hh);
                    K[e][0] = hh.kx; K[e][1] = hh.ky; K[e][2] = hh.kz; DX[e] = hh.dx;
                }
                float g[18];
#pragma unroll
                for (int v = 0; v < 18; v++) g[v] = 0.f;
#if SCAN_AFFINE
                float M1[3] = {0.f, 0.f, 0.f}, M2[3] = {0.f, 0.f, 0.f};      // sums of (pixel column offset) dp and (pixel row offset) dp; g[0..2] holds the sum of dp
#endif
                {
                float4 Sn = srd[0];
                constexpr float MC1 = FAR_N / (FAR_N - NEAR_N), MC2 = (FAR_N * NEAR_N) / (FAR_N - NEAR_N);
#pragma unroll
                for (int cb = 0; cb < 4; cb++) {
                Hit hr;
                const float pyf = sy0 + (float)cb;
                pair_planes(sx0, pyf, q0, q1, q2, hr);        // l and dy of this pixel row
#pragma unroll
                for (int ca = 0; ca < 4; ca++) {
                    const int p = 4 * cb + ca;
                    const float pxf = sx0 + (float)ca;
                    const float4 A = prow[p * 3 + 0], B = prow[p * 3 + 1], Cq = prow[p * 3 + 2];
                    const float4 S = Sn;
                    Sn = srd[(p + 1) & 15];                   // next pixel's state: read ahead of this step's write (the compiler cannot tell them apart)
                    Hit h;
                    h.kx = K[ca][0]; h.ky = K[ca][1]; h.kz = K[ca][2]; h.lx = hr.lx; h.ly = hr.ly; h.lz = hr.lz;
                    h.dx = DX[ca]; h.dy = hr.dy;
                    const bool hit = pair_intersect(Twx, Twy, Twz, opa, h);
                    const bool ok = hit & (pos <= __float_as_int(Cq.w));      // (bitwise: `&&` would put a branch into the walk)
                    // a pair that was not composited runs the same instructions with alpha = 0 and depth = 1: T x 1, X + 0
                    const float alpha = ok ? h.alpha : 0.f, depth = ok ? h.depth : 1.f;
                    const float i1a = SURFEL_RCP(1.f - alpha);
                    const float T = S.x * row_scan_mul(i1a);                  // transmittance in front of this lane's instance
                    const float w = alpha * T;
                    const float inv_d = SURFEL_RCP(depth);
                    const float mm = FMA(-(MC1 * NEAR_N), inv_d, MC1);
                    float u = FMA(mm, FMA(mm, Cq.x, Cq.y), Cq.z);
                    u = FMA(q3.w, A.x, u); u = FMA(q4.x, A.y, u); u = FMA(q4.y, A.z, u);
                    u = FMA(depth, A.w, u);
                    u = FMA(q3.x, B.x, u); u = FMA(q3.y, B.y, u); u = FMA(q3.z, B.z, u);
                    const float wu = w * u;
#if SCAN_INCL
                    const float Xn = S.y + row_scan_add_incl(wu);             // suffix sum from this lane's instance on
                    const float Xb = Xn - wu;                                 // ... and behind it
                    const float dL_dalpha = ok ? FMA(T, u, -(Xb * i1a)) : 0.f;
                    swr[p * 2] = make_float2(T, Xn);
#else
                    const float Xb = S.y + row_scan_add_excl(wu);             // suffix sum behind this lane's instance
                    const float dL_dalpha = ok ? FMA(T, u, -(Xb * i1a)) : 0.f;
                    swr[p * 2] = make_float2(T, Xb + wu);
#endif
                    if (STATS) {
                        const unsigned long long okb = __ballot(ok), vb = __ballot(valid);
                        if (lane == 0) {
                            atomicAdd(&a.stats[0], 64ull); atomicAdd(&a.stats[1], (unsigned long long)__popcll(okb));
                            atomicAdd(&a.stats[2], 1ull);
                            if (p == 0) atomicAdd(&a.stats[3], (unsigned long long)__popcll(vb));
                        }
                    }
                    float dL_dz = w * FMA(FMA(mm + mm, Cq.x, Cq.y), (MC2 * inv_d) * inv_d, A.w);
                    dL_dz += (ok & (pos == __float_as_int(S.z))) ? B.w : 0.f;
                    g[15] = FMA(w, A.x, g[15]); g[16] = FMA(w, A.y, g[16]); g[17] = FMA(w, A.z, g[17]);
                    g[11] = FMA(w, B.x, g[11]); g[12] = FMA(w, B.y, g[12]); g[13] = FMA(w, B.z, g[13]);
                    g[14] = FMA(h.G, dL_dalpha, g[14]);
                    const float nGG = -h.G * (opa * dL_dalpha);               // dL/dG * dG/drho * 2; the 0.99 clamp is pass-through
                    // low-pass branch: no gradient reaches the intersection; the selects zero (s, 1/p2) themselves (they may be inf)
                    const float sxg = h.use3d ? h.sx : 0.f, syg = h.use3d ? h.sy : 0.f, ipg = h.use3d ? h.ip : 0.f;
                    const float g2 = h.use3d ? 0.f : nGG * FILTER_INV_SQUARE;
                    const float ax = FMA(nGG, sxg, dL_dz * Twx) * ipg, ay = FMA(nGG, syg, dL_dz * Twy) * ipg;
                    const float dp2 = -FMA(ax, sxg, ay * syg);
#if SCAN_AFFINE
                    (void)pxf;
                    g[0] += ax; g[1] += ay; g[2] += dp2;
                    if (ca > 0) { M1[0] = FMA((float)ca, ax, M1[0]); M1[1] = FMA((float)ca, ay, M1[1]); M1[2] = FMA((float)ca, dp2, M1[2]); }
                    if (cb > 0) { M2[0] = FMA((float)cb, ax, M2[0]); M2[1] = FMA((float)cb, ay, M2[1]); M2[2] = FMA((float)cb, dp2, M2[2]); }
                    g[6] = FMA(dL_dz, sxg, g[6]); g[7] = FMA(dL_dz, syg, g[7]); g[8] += dL_dz;
#else
                    // -dk = dp x l ,  -dl = k x dp
                    const float nk0 = FMA(ay, h.lz, -(dp2 * h.ly)), nk1 = FMA(dp2, h.lx, -(ax * h.lz)), nk2 = FMA(ax, h.ly, -(ay * h.lx));
                    const float nl0 = FMA(h.ky, dp2, -(h.kz * ay)), nl1 = FMA(h.kz, ax, -(h.kx * dp2)), nl2 = FMA(h.kx, ay, -(h.ky * ax));
                    g[0] += nk0; g[1] += nk1; g[2] += nk2; g[3] += nl0; g[4] += nl1; g[5] += nl2;
                    g[6] = FMA(dL_dz, sxg, g[6]); g[6] = FMA(-pxf, nk0, g[6]); g[6] = FMA(-pyf, nl0, g[6]);
                    g[7] = FMA(dL_dz, syg, g[7]); g[7] = FMA(-pxf, nk1, g[7]); g[7] = FMA(-pyf, nl1, g[7]);
                    g[8] += dL_dz; g[8] = FMA(-pxf, nk2, g[8]); g[8] = FMA(-pyf, nl2, g[8]);
#endif
                    g[9] = FMA(g2, h.dx, g[9]); g[10] = FMA(g2, h.dy, g[10]);
                }
                }
                }
#if SCAN_AFFINE
                {
                    // [r6] per pair  -dk = dp x l  and  -dl = k x dp  with  k = K0 + ca Tw,  l = L0 + cb Tw  (K0, L0: the planes of the sub-tile's first
                    // column / row), so over the chunk's 16 pixels
                    //   dL/dTu = sum(dp) x L0 + sum(cb dp) x Tw ,      dL/dTv = K0 x sum(dp) + Tw x sum(ca dp) ,
                    //   dL/dTw = [the depth terms] - sx0 dL/dTu - sy0 dL/dTv - sum(ca dp) x L0 - K0 x sum(cb dp)      (the ca cb terms cancel)
                    // — 9 accumulations per pixel step instead of two cross products and 12 accumulations, 42 instructions once per chunk.
                    // Tu and Tv are read again here (kept from the top of the chunk they would occupy six registers through the walk).
                    int t2 = t;
                    asm volatile("" : "+v"(t2));
                    const float4 r0 = s_rec[t2 * 5 + 0], r1 = s_rec[t2 * 5 + 1];
                    Hit h0;
                    pair_planes(sx0, sy0, r0, make_float4(r1.x, r1.y, Twx, Twy), make_float4(Twz, 0.f, 0.f, 0.f), h0);
                    const float S0 = g[0], S1 = g[1], S2 = g[2];
                    // dL/dTu = S x L0 + M2 x Tw
                    float u0 = FMA(S1, h0.lz, -(S2 * h0.ly)), u1 = FMA(S2, h0.lx, -(S0 * h0.lz)), u2 = FMA(S0, h0.ly, -(S1 * h0.lx));
                    u0 = FMA(M2[1], Twz, u0); u0 = FMA(-M2[2], Twy, u0);
                    u1 = FMA(M2[2], Twx, u1); u1 = FMA(-M2[0], Twz, u1);
                    u2 = FMA(M2[0], Twy, u2); u2 = FMA(-M2[1], Twx, u2);
                    // dL/dTv = K0 x S + Tw x M1
                    float v0 = FMA(h0.ky, S2, -(h0.kz * S1)), v1 = FMA(h0.kz, S0, -(h0.kx * S2)), v2 = FMA(h0.kx, S1, -(h0.ky * S0));
                    v0 = FMA(Twy, M1[2], v0); v0 = FMA(-Twz, M1[1], v0);
                    v1 = FMA(Twz, M1[0], v1); v1 = FMA(-Twx, M1[2], v1);
                    v2 = FMA(Twx, M1[1], v2); v2 = FMA(-Twy, M1[0], v2);
                    // dL/dTw
                    float w0 = FMA(-sx0, u0, g[6]), w1 = FMA(-sx0, u1, g[7]), w2 = FMA(-sx0, u2, g[8]);
                    w0 = FMA(-sy0, v0, w0); w1 = FMA(-sy0, v1, w1); w2 = FMA(-sy0, v2, w2);
                    // - M1 x L0
                    w0 = FMA(-M1[1], h0.lz, w0); w0 = FMA(M1[2], h0.ly, w0);
                    w1 = FMA(-M1[2], h0.lx, w1); w1 = FMA(M1[0], h0.lz, w1);
                    w2 = FMA(-M1[0], h0.ly, w2); w2 = FMA(M1[1], h0.lx, w2);
                    // - K0 x M2
                    w0 = FMA(-h0.ky, M2[2], w0); w0 = FMA(h0.kz, M2[1], w0);
                    w1 = FMA(-h0.kz, M2[0], w1); w1 = FMA(h0.kx, M2[2], w1);
                    w2 = FMA(-h0.kx, M2[1], w2); w2 = FMA(h0.ky, M2[0], w2);
                    g[0] = u0; g[1] = u1; g[2] = u2; g[3] = v0; g[4] = v1; g[5] = v2; g[6] = w0; g[7] = w1; g[8] = w2;
                }
#endif
                s_slot[tid * 5 + 0] = make_float4(g[0], g[1], g[2], g[3]);
                s_slot[tid * 5 + 1] = make_float4(g[4], g[5], g[6], g[7]);
                s_slot[tid * 5 + 2] = make_float4(g[8], g[9], g[10], g[11]);
                s_slot[tid * 5 + 3] = make_float4(g[12], g[13], g[14], g[15]);
                s_slot[tid * 5 + 4] = make_float4(g[16], g[17], 0.f, 0.f);
            }
            __syncthreads();
            // ---- flush round c: the slot of (sub-tile s, lane r & 15) belongs to the instance of rank r = 16 c + lane on list s.
            // Fixed order: rounds ascending, sub-tiles ascending inside a round.
            if (ft < mb && c >= fcmin && c <= fcmax) {
                // (the ranks are re-read every round: kept in registers, the compiler hoists sixteen slot addresses per thread out
                // of the round loop and spills them across the walk)
                const uint4 frk = s_rank[ft];
                const uint32_t rkw[4] = {frk.x, frk.y, frk.z, frk.w};
#pragma unroll
                for (int s = 0; s < 16; s++) {
                    const int li = (int)((rkw[s >> 2] >> (8 * (s & 3))) & 0xffu) - CH * c;      // lane of the slot, if this is the rank's round
                    if ((unsigned)li < (unsigned)CH) {      // (rank 0xff = not on the list: 255 - 16 c >= 143)
                        const float4* sp = s_slot + (s * 16 + li) * 5;
                        if (fh == 0) { f0 = add4(f0, sp[0]); f1 = add4(f1, sp[1]); f2 = add4(f2, sp[2]); }
                        else { f0 = add4(f0, sp[3]); f1 = add4(f1, sp[4]); }
                    }
                }
            }
            __syncthreads();                  // slots reusable
        }
        // ---- the batch's gradient records: every staged instance gets one (zeros if no pixel took it)
        pend = ft < keep;
        if (pend) pend_slot = grec_slot(s_rec[ft * 5 + 4], tx, ty);
        step = keep;
    }
    if (pend) {
        float4* __restrict__ dst = reinterpret_cast<float4*>(a.grec + pend_slot * GREC_F);
        if (fh == 0) { dst[0] = f0; dst[1] = f1; dst[2] = f2; }
        else { dst[3] = f0; dst[4] = f1; }
    }
    finish_tail(a, range, maxc, tile, tx, ty);
}

void launch_blend_bwd_scan(const BlendBwdArgs& a, hipStream_t s) {
    const dim3 grid(a.map_len), block(BLOCK);
    const bool st = a.strm_rec != nullptr;
    if (a.stats) { if (st) hipLaunchKernelGGL((blend_bwd_scan_kernel<true, true>), grid, block, 0, s, a); else hipLaunchKernelGGL((blend_bwd_scan_kernel<true, false>), grid, block, 0, s, a); }
    else { if (st) hipLaunchKernelGGL((blend_bwd_scan_kernel<false, true>), grid, block, 0, s, a); else hipLaunchKernelGGL((blend_bwd_scan_kernel<false, false>), grid, block, 0, s, a); }
}

}  // namespace surfel
