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
    if (blockIdx.x == 0 && threadIdx.x < 64) { const uint32_t w = frame_walk(a.totals); if (threadIdx.x == 0) a.walk_word[0] = w; }      // the frame's backward walk, decided once
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

// ---------------------------------------------------------------------------------------------
// blend_fwd, software-pipelined staging ("pipe", the default; round 4).
//
// What the per-workgroup trace of the kernel above showed (profiles/r04_wg_trace.md): on a trained frame the
// slowest tile spends 41 % of its life STAGING — surfel ids -> 112-B record gather (two dependent global round trips per batch,
// issued by every resident workgroup at the same moment: the CU's fetch path, ~11 B/clk, is the bound) -> footprint test -> masks
// -> barrier — and only then walks; its VALU sits idle meanwhile (0.23 of the issue peak over the launch), and nothing else is
// resident to fill in: an object-centred frame has ~1 000 non-empty tiles, 4 per CU.  Here the records of batch b+1 travel while
// batch b is walked:
//   * batches of 128 instances, TWO record buffers in LDS, filled by LDS-DMA (global_load_lds_dwordx4: global -> LDS without a
//     VGPR, asynchronous, completion by vmcnt); the ids of the batch after that arrive the same way;
//   * wave w issues the DMA for instances [32 w, 32 w + 32) of the next batch, seven neighbouring lanes per 112-B record (whole
//     records, contiguous in LDS as in memory: a tenth of the cache lines per instruction of a quarter-per-instruction gather);
//     when it has finished its walk it waits for ITS OWN loads (no barrier), tests the footprints of those 32 instances against
//     the 16 sub-tiles (lane = instance x half of the tile) and leaves the ballots in the other mask buffer;
//   * ONE barrier per batch (the walk of batch b is over everywhere <=> buffer b is free, masks and records of b+1 are complete).
// The walk itself, and therefore every output bit, is the kernel above's (tests/test_gpu_parity.py::test_forward_kernels_are_identical).
// ---------------------------------------------------------------------------------------------
constexpr int NB = 128;                 // instances per batch
constexpr int PMSTRIDE = 4;             // a row's four mask words: one aligned 16-B read

// sub-tile bits (id = 4 * by + bx) of the two 4-row strips by = 2 h, 2 h + 1
__device__ __forceinline__ unsigned subtile_overlap_half(const Foot& f, int tile_x0, int tile_y0, int h) {
    unsigned ov = 0;
#pragma unroll
    for (int b = 0; b < 2; b++) {
        const int by = 2 * h + b;
        float xmin, xmax;
        foot_strip(f, (float)(tile_y0 + 4 * by), (float)(tile_y0 + 4 * by + 3), xmin, xmax);
        xmin -= (float)tile_x0; xmax -= (float)tile_x0;
#pragma unroll
        for (int bx = 0; bx < 4; bx++) ov |= (xmin <= (float)(4 * bx + 3) && xmax >= (float)(4 * bx)) ? (1u << (4 * b + bx)) : 0u;
    }
    return ov;
}

static_assert(REC_Q == 7, "the DMA's piece -> record arithmetic assumes 112-B records");
template <bool STATS>
__global__ void __launch_bounds__(BLOCK, 5) blend_fwd_pipe_kernel(BlendFwdArgs a) {
    __shared__ float4 s_rec[2][NB][REC_Q];             // 28 KB: the whole 112-B records of two batches, [buffer][instance][quarter]
    __shared__ __attribute__((aligned(16))) uint32_t s_mask[2][16 * PMSTRIDE];     // [buffer][sub-tile][word]
    __shared__ int s_alldone[2][4];
    __shared__ uint32_t s_list[4][4][NB / 4 + 1];      // 2.1 KB: per wave and DPP row, the staged indices (bytes) of the row's visits of this batch
    __shared__ uint32_t s_ids[4][32];                  // per wave, the surfel ids of its 32 instances of the batch after next (DMA as well:
                                                       // a load hipcc tracks would make it wait for vmcnt(0) — i.e. for the record DMA — at its next use)
    if (blockIdx.x == 0 && threadIdx.x < 64) { const uint32_t w = frame_walk(a.totals); if (threadIdx.x == 0) a.walk_word[0] = w; }      // the frame's backward walk, decided once
    const int tile = block_tile(a.tile_map, a.map_flag, blockIdx.x, a.gx * a.gy);
    if (tile < 0) return;
    const int tx = tile % a.gx, ty = tile / a.gx;
    int lx, ly, sub;
    thread_pixel(threadIdx.x, lx, ly, sub);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int pxi = tx * TILE + lx, pyi = ty * TILE + ly;
    const bool inside = pxi < a.W && pyi < a.H;
    const float pxf = (float)pxi, pyf = (float)pyi;
    const uint2 range_v = a.ranges[tile];
    const uint2 range = make_uint2(__builtin_amdgcn_readfirstlane(range_v.x), __builtin_amdgcn_readfirstlane(range_v.y));      // (uniform: kept in SGPRs)
    const int n = (int)(range.y - range.x);

    bool done = !inside;
    float T = 1.f, C0 = 0.f, C1 = 0.f, C2 = 0.f, N0 = 0.f, N1 = 0.f, N2 = 0.f;
    float D = 0.f, M1 = 0.f, M2 = 0.f, dist = 0.f, med = 0.f;
    uint32_t last = 0, medc = 0;
    unsigned npairs = 0;
    constexpr float MC1 = FAR_N / (FAR_N - NEAR_N);
    for (int k = lane; k < 4 * (NB / 4 + 1); k += 64) (&s_list[wave][0][0])[k] = 0u;      // (bytes past a list's end are read as indices: any valid one will do)
    const char* const recb = reinterpret_cast<const char*>(a.rec);
    // This wave's share of a batch's DMA: the records of instances [32 wave, 32 wave + 32) of the batch that starts at list position
    // `base` — nine records (63 sixteen-byte pieces) per wave-wide DMA instruction, four instructions:
    // seven neighbouring lanes fetch one record's 112 contiguous bytes (one or two cache lines per record and instruction, ~10 - 18
    // lines per instruction) and the pieces land in LDS in the same order (the DMA writes base + 16 * lane), i.e. as whole records.
    // (One record quarter per instruction — lane = record — touched 64 lines per instruction and each line up to seven times.)
    const unsigned rec_base = lds_offset(&s_rec[0][32 * wave][0]);
    const int dma_r = (lane * 9363) >> 16, dma_off = 16 * (lane - REC_Q * dma_r);      // lane / 7 (exact below 224), byte offset of the lane's piece
    auto issue = [&](int base, int buf) {
        const unsigned dst = __builtin_amdgcn_readfirstlane(rec_base + (unsigned)(NB * REC_Q * 16) * (unsigned)buf);
#pragma unroll
        for (int i = 0; i < 4; i++) {      // instruction i: records 9 i .. 9 i + 8 of the wave's 32, lanes 0 .. 62
            const int r = 9 * i + dma_r;
            if (lane < 63 && r < 32 && base + 32 * wave + r < n) {
                const uint32_t id = s_ids[wave][r];
                dma16(recb + (size_t)id * (REC_F * 4) + dma_off, dst + (unsigned)(9 * REC_Q * 16) * (unsigned)i);
            }
        }
    };
    // ... and, once they have landed, of its masks: lane = (instance, half of the tile): the sub-tiles of pixel rows 8 half .. 8 half + 7
    // x its 32 instances -> word `wave` of those sub-tiles' masks
    auto masks = [&](int base, int buf) {
        unsigned ov = 0;
        const int j = 32 * wave + (lane & 31), half = lane >> 5;
        // the tile stream (surfel_common.h): what blend_bwd will need of these 32 instances, left behind in list order by the wave that
        // holds it — 2.5 KB of records as three coalesced 16-B stores per lane (read from LDS behind the ballots: pieces requested earlier —
        // under the footprint arithmetic or under the ballots — measured the same or cost a wave per SIMD), and the 16 footprint bits per instance
        const int first = base + 32 * wave;                                  // list position (0-based) of this wave's first instance
        const int npc = STRM_Q * min(32, n - first);                         // 16-B pieces this wave owes the stream
        if (base + j < n) ov = subtile_overlap_half(make_foot(s_rec[buf][j][2], s_rec[buf][j][5], s_rec[buf][j][6]), tx * TILE, ty * TILE, half);
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const unsigned long long b = __ballot((ov >> k) & 1u);
            if (lane == 0) { s_mask[buf][k * PMSTRIDE + wave] = (uint32_t)b; s_mask[buf][(8 + k) * PMSTRIDE + wave] = (uint32_t)(b >> 32); }
        }
        if (a.strm_rec) {
            const unsigned other = (unsigned)__shfl_xor((int)ov, 32);      // the other half of the tile
            // (addresses computed HERE, from a lane id the compiler cannot see through: hoisted out of the batch loop they occupy registers
            // through the walk)
            unsigned ln = (unsigned)lane;
            asm volatile("" : "+v"(ln));
            const unsigned rx = __builtin_amdgcn_readfirstlane(range.x);
            if (ln < 32u && first + (int)ln < n) a.strm_mask[(size_t)rx + (size_t)(first + (int)ln)] = subtile_bits_to_rows(ov | (other << 8));
            float4* __restrict__ dst = a.strm_rec + ((size_t)rx + (size_t)first) * STRM_Q + ln;
            const float4* src = &s_rec[buf][32 * wave][0];
            const int p0 = (int)ln, p1 = (int)ln + 64, p2 = (int)ln + 128;           // piece p = quarter p % 5 of instance p / 5
            const int j0 = (p0 * 13108) >> 16, j1 = (p1 * 13108) >> 16, j2 = (p2 * 13108) >> 16;      // p / 5 (exact below 65536 / 4)
            if (p0 < npc) dst[0] = src[REC_Q * j0 + (p0 - STRM_Q * j0)];
            if (p1 < npc) dst[64] = src[REC_Q * j1 + (p1 - STRM_Q * j1)];
            if (p2 < npc) dst[128] = src[REC_Q * j2 + (p2 - STRM_Q * j2)];
        }
    };
    const unsigned ids_base = __builtin_amdgcn_readfirstlane(lds_offset(&s_ids[wave][0]));
    auto issue_ids = [&](int base) {      // ids of this wave's instances of the batch at `base` -> s_ids[wave] (consumed by issue() before the next ones are requested)
        const int k = base + 32 * wave + lane;
        if (lane < 32 && k < n) dma4(a.point_list + range.x + k, ids_base);
    };

    if (n > 0) {
        {
            const int k = 32 * wave + lane;      // (the one load hipcc tracks: waited for before any DMA is issued)
            if (lane < 32) s_ids[wave][lane] = k < n ? a.point_list[range.x + k] : 0u;
        }
        issue(0, 0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // (the ids are in registers before the DMA below may overwrite them)
        issue_ids(NB);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        masks(0, 0);
        __syncthreads();
        for (int base = 0, buf = 0; base < n; base += NB, buf ^= 1) {
            const bool more = base + NB < n;
            if (more) {      // next batch's records (their ids landed with this batch's records), and the ids of the batch behind it
                issue(base + NB, buf ^ 1);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                issue_ids(base + 2 * NB);
            }
            // ---- the walk.  On frames with few resident waves the bound is the wave's own issue rate plus the LDS round trips it
            // waits for (SQ counters of the batch-synchronous kernel on a trained frame: 40 % of the wave-cycles issuing, 18 % stalled
            // on issue, 42 % parked at s_waitcnt / barriers).  So: (1) each row's mask is expanded ONCE per batch into a byte list of
            // staged indices, by the row's own lanes — a visit then costs a byte extract instead of find-first-set / clear / word
            // refill / two votes, and the loop is a counted loop (trip count = the longest of the wave's four lists); (2) the record of
            // visit v + 1 is read from LDS while visit v is being computed.  Visit order per row = list order = mask-bit order: the
            // results are the mask walk's, bit for bit.
            const float4* R = &s_rec[buf][0][0];      // R[REC_Q * j + quarter]
            uint32_t* const lrow = &s_list[wave][lane >> 4][0];
            int len;
            {
                const int i16 = lane & 15;
                const uint4 mw = *reinterpret_cast<const uint4*>(&s_mask[buf][sub * PMSTRIDE]);
                const unsigned long long db = __ballot(done);      // a row whose 16 pixels are all saturated skips the batch
                const bool rowdone = (((unsigned)(db >> (lane & 48))) & 0xffffu) == 0xffffu;
                len = rowdone ? 0 : (int)(__popc(mw.x) + __popc(mw.y) + __popc(mw.z) + __popc(mw.w));
                const uint32_t wsel = (i16 & 8) ? ((i16 & 4) ? mw.w : mw.z) : ((i16 & 4) ? mw.y : mw.x);
                const uint32_t byte = (wsel >> (8 * (i16 & 3))) & 0xffu;      // instances 8 i16 .. 8 i16 + 7 of the row's mask
                int off = __popc(byte);                                    // exclusive prefix over the row's 16 lanes
                int t;
                t = __builtin_amdgcn_update_dpp(0, off, 0x111, 0xf, 0xf, true); off += t;      // row_shr:1
                t = __builtin_amdgcn_update_dpp(0, off, 0x112, 0xf, 0xf, true); off += t;      // row_shr:2
                t = __builtin_amdgcn_update_dpp(0, off, 0x114, 0xf, 0xf, true); off += t;      // row_shr:4
                t = __builtin_amdgcn_update_dpp(0, off, 0x118, 0xf, 0xf, true); off += t;      // row_shr:8
                off -= __popc(byte);
                uint8_t* const lb = reinterpret_cast<uint8_t*>(lrow);
#pragma unroll
                for (int k = 0; k < 8; k++)
                    if ((byte >> k) & 1u) lb[off + __popc(byte & ((1u << k) - 1u))] = (uint8_t)(8 * i16 + k);
            }
            int niter = max(max(__builtin_amdgcn_readlane(len, 0), __builtin_amdgcn_readlane(len, 16)),
                            max(__builtin_amdgcn_readlane(len, 32), __builtin_amdgcn_readlane(len, 48)));
            if (niter > 0) {
                uint32_t jw = lrow[0];
                int j = (int)(jw & 0xffu);
                float4 q0 = R[REC_Q * j], q1 = R[REC_Q * j + 1], q2 = R[REC_Q * j + 2], q3 = R[REC_Q * j + 3];
                float2 q4 = *reinterpret_cast<const float2*>(&R[REC_Q * j + 4]);
                for (int i = 0; i < niter; i += 4) {
                    const uint32_t jw_next = lrow[(i >> 2) + 1];      // (one word past the longest list: allocated, never used)
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        if (i + k >= niter) break;
                        const bool act = i + k < len;
                        // the next visit's record: on its way while this one is computed
                        const int jn = (int)(k < 3 ? (jw >> (8 * (k + 1))) & 0xffu : jw_next & 0xffu);
                        const float4 n0 = R[REC_Q * jn], n1 = R[REC_Q * jn + 1], n2 = R[REC_Q * jn + 2], n3 = R[REC_Q * jn + 3];
                        const float2 n4 = *reinterpret_cast<const float2*>(&R[REC_Q * jn + 4]);
                        Hit hh;
                        const bool hit = pair_hit(pxf, pyf, q0, q1, q2, hh);
                        const float depth = hh.depth, alpha = hh.alpha;
                        const bool ok = act & (!done) & hit;
                        const float testT = T * (1.f - alpha);
                        const bool term = ok & (testT < T_EPS);      // the terminating surfel is not composited
                        done |= term;
                        if (ok & !term) {
                            if (STATS) npairs++;
                            const uint32_t contributor = (uint32_t)(base + j + 1);
                            const float w_ = alpha * T;
                            const float mm = MC1 - (MC1 * NEAR_N) * SURFEL_RCP(depth);
                            dist += (mm * (mm * (1.f - T) - 2.f * M1) + M2) * w_;
                            D += depth * w_;
                            M1 += mm * w_;
                            M2 += mm * mm * w_;
                            if (T > 0.5f) { med = depth; medc = contributor; }
                            N0 += q3.x * w_; N1 += q3.y * w_; N2 += q3.z * w_;
                            C0 += q3.w * w_; C1 += q4.x * w_; C2 += q4.y * w_;
                            T = testT;
                            last = contributor;
                        }
                        q0 = n0; q1 = n1; q2 = n2; q3 = n3; q4 = n4; j = jn;
                    }
                    jw = jw_next;
                    if (__all(done)) break;
                }
            }
            if (more) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's own DMA of the next batch
                masks(base + NB, buf ^ 1);
            }
            const int wave_done = __all(done) ? 1 : 0;      // (evaluated by the whole wave, not under the lane-0 branch)
            if (lane == 0) s_alldone[buf][wave] = wave_done;
            __syncthreads();
            if ((s_alldone[buf][0] & s_alldone[buf][1] & s_alldone[buf][2] & s_alldone[buf][3]) != 0) break;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (a DMA issued for a batch that saturation made unnecessary must land before the LDS is handed back)
    }
    if (STATS) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) npairs += __shfl_xor(npairs, o);
        if (lane == 0) atomicAdd(&a.stats[6], (unsigned long long)npairs);
    }
    // (the pixel's coordinates are derived again here, from a thread id the compiler cannot see through: kept from the top of the kernel
    // they occupy two registers through the walk — the 97th and 98th)
    unsigned tid_e = threadIdx.x;
    asm volatile("" : "+v"(tid_e));
    int lx_e, ly_e, sub_e;
    thread_pixel((int)tid_e, lx_e, ly_e, sub_e);
    const int pxe = tx * TILE + lx_e, pye = ty * TILE + ly_e;
    if (pxe < a.W && pye < a.H) {
        const size_t HW = (size_t)a.H * a.W;
        const size_t pix = (size_t)pye * a.W + pxe;
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

static int g_fwd_pipe = 1;      // surfel_set_option("fwd_pipe", .): 1 pipelined staging (default), 0 the batch-synchronous kernel
void set_fwd_pipe(int v) { g_fwd_pipe = v != 0; }

bool launch_blend_fwd_writes_stream(const BlendFwdArgs& a) { return a.strm_rec != nullptr && g_fwd_pipe && !(a.avg_list > 2048); }

void launch_blend_fwd(const BlendFwdArgs& a, hipStream_t s) {
    // Crowded frames (thousands of instances per tile of which a few per cent are ever staged: C5, 10 M surfels at 4K) end most tiles
    // inside their second batch: the prefetch of a batch nobody walks and a barrier per 128 instead of 256 instances cost the pipelined
    // kernel 17 % there (1.27 vs 1.08 ms) — such frames keep the batch-synchronous kernel.  Same bits either way.
    const bool crowded = a.avg_list > 2048;
    if (g_fwd_pipe && !crowded) {
        if (a.stats) hipLaunchKernelGGL(blend_fwd_pipe_kernel<true>, dim3(a.map_len), dim3(BLOCK), 0, s, a);
        else hipLaunchKernelGGL(blend_fwd_pipe_kernel<false>, dim3(a.map_len), dim3(BLOCK), 0, s, a);
        return;
    }
    if (a.stats) hipLaunchKernelGGL(blend_fwd_kernel<true>, dim3(a.map_len), dim3(BLOCK), 0, s, a);
    else hipLaunchKernelGGL(blend_fwd_kernel<false>, dim3(a.map_len), dim3(BLOCK), 0, s, a);
}

}  // namespace surfel
