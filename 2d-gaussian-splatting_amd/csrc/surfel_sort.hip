// surfel_sort.hip — stable LSD radix sort of (u32 key, u32 value) pairs for gfx950: ONE launch per 8-bit pass.
//
// Why not rocPRIM here: the two sorts on the hot path are small and oddly shaped — P surfels by 32 depth bits and R
// instances by the 12-17 tile-id bits — and at BASELINE configs[1] (300 k surfels / 0.6 M instances) every extra
// launch costs more than the bytes it moves.  Structure ("onesweep"):
//   os_hist   : digit histograms of ALL passes in one sweep over the keys (the multiset of digits of a pass does not
//               depend on the order earlier passes leave behind); also clears the look-back state.  Callers that
//               produce the keys themselves can fill the histogram on the fly and skip this launch.
//   os_pass   : one launch per pass.  A 1024-thread workgroup takes the next 4096-item tile (ticket order = input order), ranks
//               its items with wave64 match-by-ballot (no atomics, order-preserving => stable), publishes its digit
//               counts, and obtains the counts of all earlier tiles by decoupled look-back over their published
//               (aggregate | inclusive-prefix) words — one 32-bit word per (tile, digit) carrying flag and value, stored
//               and loaded at agent scope, so there is no separate payload to order.  Predecessor tiles hold smaller
//               tickets, i.e. they are already running, so the spin always terminates.
// The tile's items pass through LDS in tile-local sorted order and leave as same-digit runs written by consecutive lanes.
// Traffic per pass: 16 B/item + 1 KB of status per tile.
// The look-back form is used while the whole grid is (nearly) resident (n <= RS_ONESWEEP_MAX): there a pass costs
// 16 us instead of 30 us for three launches.  Beyond that every tile is resident and publishing at once, a tile's look-back reads
// O(tiles) words per digit and the chain throttles the scatter (measured at 2.2 M and 10 M items, both rounds 2 and 6), so
// large inputs take three launches per pass:
//   rs_hist_fat (digit counts per tile) -> rs_scan (one workgroup per digit over the tiles) -> the same tile body (os_pass_fat<false>).
// [r6] Until round 6 the large path scattered from registers (2048-item tiles, a digit's 8 items per tile stored by whichever lanes
// ranked them: 64 lines per store instruction, 1.2 TB/s per pass) and lost to rocPRIM on tile-id sorts; through LDS it reaches
// 2.0 TB/s per pass at 8 M items (profiles/r06_sort_ab.jsonl: C4 depth sort + scan 196 -> 157 us, tile sort 169 (rocPRIM) -> 130 us).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>

#include "surfel_common.h"
#include "surfel_kernels.h"

namespace surfel {

constexpr size_t RS_ONESWEEP_MAX_ITEMS = (size_t)1 << 20;
// largest input the one-launch-per-pass look-back passes take (above: histogram + scan + scatter per pass, or rocPRIM: use_rocprim).
// [r6] raised to 2^22 for the 2.2 M-key depth sort of the garden state it LOST: 0.220 -> 0.248 ms (profiles/r06_ab_preprocess_dma.jsonl)
constexpr size_t g_onesweep_max = RS_ONESWEEP_MAX_ITEMS;
// Large inputs (n > RS_ONESWEEP_MAX: the P-sized depth sort at C4 / C5 and the R-sized tile sort of 1e7 - 1e8 instances) can be
// handed to rocprim::radix_sort_pairs (the north star names rocPRIM for the global key sort) instead of the three-launch path
// below: surfel_set_option("large_sort", 1).  Both are stable LSD sorts on [begin_bit, end_bit): results are identical.
int g_large_sort_impl = 2;      // 0 own, 1 rocPRIM, 2 auto
void set_large_sort_impl(int v) { g_large_sort_impl = v < 0 ? 0 : (v > 3 ? 2 : v); }      // 3: rocPRIM for every size (measurement only)
// auto (measured on MI355X, profiles/r06_sort_ab.jsonl): the library's own passes up to 2^25 items (8.1 M tile-id pairs: 130 vs
// 168 us; 2 M / 10 M depth keys: 0.157 vs 0.216 / 0.584 vs 0.651 ms incl. the scan), rocPRIM above (1.3e8 tile-id pairs: 2.04 vs
// 2.44 ms — at that size 8192-item tiles draw level, 2.02, but lose below).
static inline bool use_rocprim(size_t n, int begin_bit, int end_bit) {
    // key fields that do not start at bit 0 always take the library's own passes: rocprim::radix_sort_pairs of this ROCm returned
    // wrongly ordered values for [24, 32), [30, 32) and [31, 32) at >= 4096 items (scripts/dbg/rocprim_small.py; fields starting at
    // bit 0 — all the hot path sorts on — are exact at every size tested)
    if (begin_bit != 0) return false;
    if (g_large_sort_impl == 3) return true;
    if (n <= g_onesweep_max) return false;
    if (g_large_sort_impl != 2) return g_large_sort_impl == 1;
    return n > ((size_t)1 << 25);
}
static size_t rocprim_sort_temp_bytes(size_t n) {
    // (queried for the full 32-bit key: an upper bound for every [begin_bit, end_bit) the library sorts on; the host-side query is
    // cached per item count — frames repeat their sizes, and this sits on the forward's launch path)
    thread_local size_t last_n = 0, last_bytes = 0;
    if (n == last_n && last_bytes) return last_bytes;
    size_t bytes = 0;
    uint32_t* nul = nullptr;
    (void)rocprim::radix_sort_pairs(nullptr, bytes, nul, nul, nul, nul, n, 0u, 32u, (hipStream_t)0);
    last_n = n; last_bytes = bytes;
    return bytes;
}

constexpr int RS_THREADS = 256;
// items per thread: 8 (2048-item tiles) for large inputs — long same-digit store runs, 1 KB of status per 16 KB of
// pairs; 2 (512-item tiles) for small ones, where the whole grid is resident and the critical path of one tile
// (sweeps x ballot ranking at one wave per SIMD) is what a pass costs.
// (512-item tiles were measured 2x SLOWER at 0.3-0.6 M items: 4x the tickets, status words and look-back depth.)
constexpr int RS_IPT = 8;
constexpr int RS_TILE = RS_THREADS * RS_IPT;
static inline bool rs_onesweep(size_t n) { return n <= g_onesweep_max; }
constexpr int RS_BITS = 8;
constexpr int RS_RADIX = 1 << RS_BITS;
constexpr int RS_MAX_PASSES = 4;
constexpr uint32_t ST_AGG = 1u << 30, ST_PREFIX = 2u << 30, ST_VALUE = (1u << 30) - 1u;

// scratch layout (u32 words): ghist[RS_MAX_PASSES][256] | ticket[RS_MAX_PASSES] (+pad to 64) | status[passes][nblocks][256]
constexpr size_t RS_HEAD_WORDS = RS_MAX_PASSES * RS_RADIX + 64;

static inline uint32_t rs_nblocks(size_t n) { return (uint32_t)((n + RS_TILE - 1) / RS_TILE); }

size_t radix_sort_scratch_bytes(size_t n) {
    if (!rs_onesweep(n)) {
        const size_t own = ((size_t)RS_RADIX * rs_nblocks(n) + RS_RADIX) * sizeof(uint32_t) + 256;   // hist[digit][tile] | total[digit]
        const size_t rp = rocprim_sort_temp_bytes(n) + 256;
        return own > rp ? own : rp;
    }
    const size_t own = (RS_HEAD_WORDS + (size_t)RS_MAX_PASSES * rs_nblocks(n) * RS_RADIX) * sizeof(uint32_t) + 256;
    const size_t rp = g_large_sort_impl == 3 ? rocprim_sort_temp_bytes(n) + 256 : 0;
    return own > rp ? own : rp;
}
int radix_sort_passes(size_t, int begin_bit, int end_bit) { return (end_bit - begin_bit + RS_BITS - 1) / RS_BITS; }
// which buffer pair radix_sort_pairs_u32 will leave the result in (0: a, 1: b) — callers that need it in a particular buffer
// assign a / b accordingly BEFORE the call
int radix_sort_result_buffer(size_t n, int begin_bit, int end_bit) {
    if (n == 0 || end_bit <= begin_bit) return 0;
    if (use_rocprim(n, begin_bit, end_bit)) return 1;
    return radix_sort_passes(n, begin_bit, end_bit) & 1;
}
size_t radix_sort_head_bytes() { return RS_HEAD_WORDS * sizeof(uint32_t); }

__device__ __forceinline__ uint32_t ld_agent(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_agent(uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// digit histograms of every pass -> ghist (pre-zeroed); clears the look-back status words of this sort
__global__ void __launch_bounds__(RS_THREADS) os_hist_kernel(const uint32_t* __restrict__ keys, uint32_t n, int begin_bit, int end_bit,
                                                             int passes, uint32_t* __restrict__ ghist, uint32_t* __restrict__ status,
                                                             uint32_t nblocks) {
    __shared__ uint32_t s_h[RS_MAX_PASSES][RS_RADIX];
    for (int p = 0; p < passes; p++) s_h[p][threadIdx.x] = 0;
    for (int p = 0; p < passes; p++) status[((size_t)p * nblocks + blockIdx.x) * RS_RADIX + threadIdx.x] = 0u;
    __syncthreads();
    const uint32_t base = blockIdx.x * (RS_THREADS * RS_IPT);
#pragma unroll
    for (int it = 0; it < RS_IPT; it++) {
        const uint32_t e = base + it * RS_THREADS + threadIdx.x;
        if (e < n) {
            const uint32_t k = keys[e];
            for (int p = 0; p < passes; p++) {
                const int bit = begin_bit + p * RS_BITS;
                const int nb = min(RS_BITS, end_bit - bit);
                atomicAdd(&s_h[p][(k >> bit) & ((1u << nb) - 1u)], 1u);
            }
        }
    }
    __syncthreads();
    for (int p = 0; p < passes; p++) {
        const uint32_t c = s_h[p][threadIdx.x];
        if (c) atomicAdd(&ghist[p * RS_RADIX + threadIdx.x], c);
    }
}

// ---- large-n path: digit counts per 4096-item tile -> hist[digit][tile] (rs_hist_fat_kernel), then one workgroup per digit scans its row
constexpr int FTH_THREADS = 1024;
template <int FTH_IPT>
__global__ void __launch_bounds__(FTH_THREADS) rs_hist_fat_kernel(const uint32_t* __restrict__ keys, uint32_t n, int shift, uint32_t mask,
                                                                  uint32_t* __restrict__ hist, uint32_t nblocks) {
    __shared__ uint32_t s_h[RS_RADIX];
    if (threadIdx.x < RS_RADIX) s_h[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t base = blockIdx.x * (uint32_t)(FTH_THREADS * FTH_IPT);
#pragma unroll
    for (int it = 0; it < FTH_IPT; it++) {
        const uint32_t e = base + it * FTH_THREADS + threadIdx.x;
        if (e < n) atomicAdd(&s_h[(keys[e] >> shift) & mask], 1u);
    }
    __syncthreads();
    if (threadIdx.x < RS_RADIX) hist[(size_t)threadIdx.x * nblocks + blockIdx.x] = s_h[threadIdx.x];
}
__global__ void __launch_bounds__(RS_THREADS) rs_scan_kernel(uint32_t* __restrict__ hist, uint32_t nblocks, uint32_t* __restrict__ total) {
    __shared__ uint32_t s_w[4];
    __shared__ uint32_t s_carry;
    uint32_t* row = hist + (size_t)blockIdx.x * nblocks;
    if (threadIdx.x == 0) s_carry = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (uint32_t c0 = 0; c0 < nblocks; c0 += RS_THREADS) {
        const uint32_t i = c0 + threadIdx.x;
        const uint32_t v = i < nblocks ? row[i] : 0u;
        uint32_t x = v;                           // inclusive wave scan
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const uint32_t y = __shfl_up(x, o); if (lane >= o) x += y; }
        if (lane == 63) s_w[wave] = x;
        __syncthreads();
        uint32_t woff = 0;
        for (int w = 0; w < wave; w++) woff += s_w[w];
        const uint32_t carry = s_carry;
        if (i < nblocks) row[i] = carry + woff + x - v;
        __syncthreads();
        if (threadIdx.x == RS_THREADS - 1) s_carry = carry + woff + x;
        __syncthreads();
    }
    if (threadIdx.x == 0) total[blockIdx.x] = s_carry;
}

// ---- the look-back pass for <= 2^20 items, fat tiles -------------------------------------------------------------------------
// Same algorithm as os_pass_kernel<true>, 1024-thread workgroups with 4096-item tiles (8192 until round 5), and the tile's items go through LDS in
// tile-local sorted order before they are stored.  Measured on the thin (2048-item) pass at 0.5 M items (r03g_C2_kernel_stats):
// 18.5 us for the 4-bit digit, 30.5 us for the 8-bit digit — the difference is the scatter: a thin tile holds 8 items per 8-bit
// digit, stored by whichever lane ranked them (64 different lines per store instruction).  Here a digit's run in a tile is
// 16 items = 64 B, written by consecutive lanes; and the look-back chain is half as long.
// (Round 5: 1024 x 4 = 4096-item tiles — 128 workgroups for an 800x800 frame's 0.5 M pairs instead of 64 — measured against 1024 x 8,
// 1024 x 3, 1024 x 2, 768 x 4, 512 x 16, 512 x 8, 512 x 4: 36.4 us for the two passes against 39.0 / 38.2 / 42.7 / 38.3 / 41.0 / 37.5 / 45.1.)
constexpr int FT_THREADS = 1024, FT_IPT = 4, FT_TILE = FT_THREADS * FT_IPT, FT_WAVES = FT_THREADS / 64;
constexpr int FT_LB = 32;      // look-back window: at most 256 tiles exist, all resident and publishing at about the same moment
static inline uint32_t ft_nblocks(size_t n) { return (uint32_t)((n + FT_TILE - 1) / FT_TILE); }

// LOOKBACK = false (inputs above 2^20 items, round 6): the same tile body behind the histogram + scan launches of the large-n path
// (ghist = total[digit] from rs_scan, status = hist[digit][tile] exclusive-scanned over the 4096-item tiles, tile = blockIdx.x) — the
// thin pass stores a digit's 8 items per tile from whichever lanes ranked them; here they leave LDS as 64-B runs.
template <bool LOOKBACK, int IPT>
__global__ void __launch_bounds__(FT_THREADS) os_pass_fat_kernel(const uint32_t* __restrict__ keys, const uint32_t* __restrict__ vals,
                                                                 uint32_t* __restrict__ keys_out, uint32_t* __restrict__ vals_out, uint32_t n,
                                                                 int shift, uint32_t mask, const uint32_t* __restrict__ ghist,
                                                                 uint32_t* __restrict__ status, uint32_t* __restrict__ ticket,
                                                                 const uint32_t* __restrict__ n_dev, int hstride, uint2* __restrict__ ranges,
                                                                 uint32_t nblocks) {
    __shared__ uint32_t s_cnt[FT_WAVES][RS_RADIX];      // per-wave digit counts -> then per-wave tile-local offsets
    constexpr int TILE_ITEMS = FT_THREADS * IPT;
    __shared__ uint2 s_item[TILE_ITEMS];                   // (key, value) in tile-local sorted order
    __shared__ uint32_t s_gbase[RS_RADIX];              // global slot of a digit's first item of this tile, minus its tile-local slot
    __shared__ uint32_t s_w[4][2];
    __shared__ uint32_t s_tile;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t d_t = threadIdx.x;                   // (threads < 256) the digit this thread owns in the offset phases
    if (LOOKBACK && threadIdx.x == 0) s_tile = atomicAdd(ticket, 1u);
    for (int i = threadIdx.x; i < FT_WAVES * RS_RADIX; i += FT_THREADS) (&s_cnt[0][0])[i] = 0u;
    __syncthreads();
    const uint32_t tile = LOOKBACK ? s_tile : blockIdx.x;
    if (n_dev) n = min(n, *n_dev);
    if (tile * (uint32_t)TILE_ITEMS >= n) return;
    const uint32_t wbase = tile * (uint32_t)TILE_ITEMS + wave * (64 * IPT);
    uint32_t k[IPT], v[IPT], rank[IPT];
#pragma unroll
    for (int it = 0; it < IPT; it++) {
        const uint32_t e = wbase + it * 64 + lane;
        const bool valid = e < n;
        k[it] = valid ? keys[e] : 0xffffffffu;
        v[it] = valid ? vals[e] : 0u;
    }
#pragma unroll
    for (int it = 0; it < IPT; it++) {
        const uint32_t e = wbase + it * 64 + lane;
        const bool valid = e < n;
        const uint32_t d = (k[it] >> shift) & mask;
        unsigned long long mm = __ballot(valid);
#pragma unroll
        for (int b = 0; b < RS_BITS; b++) {
            const bool bit = (d >> b) & 1u;
            const unsigned long long bal = __ballot(bit);
            mm &= bit ? bal : ~bal;
        }
        if (!valid) mm = 0ull;
        const uint32_t below = __builtin_amdgcn_mbcnt_hi((uint32_t)(mm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mm, 0u));
        const uint32_t prev = valid ? s_cnt[wave][d] : 0u;   // same-wave LDS ops are program-ordered
        rank[it] = prev + below;
        if (valid && below == 0) s_cnt[wave][d] = prev + (uint32_t)__popcll(mm);
    }
    __syncthreads();
    uint32_t cw[FT_WAVES], cnt = 0, excl = 0, tot = 0, x = 0, y = 0;      // (threads < 256: one digit each)
    if (threadIdx.x < RS_RADIX) {
#pragma unroll
        for (int w = 0; w < FT_WAVES; w++) { cw[w] = s_cnt[w][d_t]; cnt += cw[w]; }
        uint32_t* my = status + (size_t)tile * RS_RADIX + d_t;
        if (!LOOKBACK) {
            excl = status[(size_t)d_t * nblocks + tile];
        } else if (tile == 0) {
            st_agent(my, cnt | ST_PREFIX);
        } else {
            st_agent(my, cnt | ST_AGG);
            int p = (int)tile - 1;
            bool done = false;
            // Look back FT_LB tiles per round with independent loads.  When the whole grid is resident every tile publishes its
            // aggregate at the same moment and a one-word-per-step walk would resolve in ~sqrt(2 * tiles) dependent agent-scope loads
            // (~0.7 us each); batching the window cuts the number of dependent rounds.  A not-yet-published word stops the round; the
            // next round re-reads from it.
            while (!done) {
                uint32_t w[FT_LB];
#pragma unroll
                for (int i = 0; i < FT_LB; i++) w[i] = (p - i >= 0) ? ld_agent(status + (size_t)(p - i) * RS_RADIX + d_t) : ST_PREFIX;
                int used = 0;
#pragma unroll
                for (int i = 0; i < FT_LB; i++) {
                    if (!done && used == i) {
                        if ((w[i] >> 30) != 0u) {
                            excl += w[i] & ST_VALUE;
                            used = i + 1;
                            if (w[i] & ST_PREFIX) done = true;
                        }
                    }
                }
                p -= used;
                if (!done && used == 0) __builtin_amdgcn_s_sleep(1);
            }
            st_agent(my, (excl + cnt) | ST_PREFIX);
        }
        // exclusive scans over the digits: of the digit totals (global start of a digit) and of this tile's counts (tile-local start)
        tot = ghist[(size_t)d_t * hstride];
        x = tot; y = cnt;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t xa = __shfl_up(x, o), ya = __shfl_up(y, o);
            if (lane >= o) { x += xa; y += ya; }
        }
        if (lane == 63) { s_w[wave][0] = x; s_w[wave][1] = y; }
    }
    __syncthreads();
    if (threadIdx.x < RS_RADIX) {
        uint32_t goff = x - tot + excl, loff = y - cnt;
        for (int w = 0; w < wave; w++) { goff += s_w[w][0]; loff += s_w[w][1]; }
        s_gbase[d_t] = goff - loff;
        uint32_t run = loff;
#pragma unroll
        for (int w = 0; w < FT_WAVES; w++) { s_cnt[w][d_t] = run; run += cw[w]; }
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < IPT; it++) {
        const uint32_t e = wbase + it * 64 + lane;
        if (e < n) {
            const uint32_t d = (k[it] >> shift) & mask;
            s_item[s_cnt[wave][d] + rank[it]] = make_uint2(k[it], v[it]);
        }
    }
    __syncthreads();
    const uint32_t items = min((uint32_t)TILE_ITEMS, n - tile * (uint32_t)TILE_ITEMS);
#pragma unroll
    for (int it = 0; it < IPT; it++) {
        const uint32_t q = (uint32_t)it * FT_THREADS + threadIdx.x;
        if (q < items) {
            const uint2 kv = s_item[q];
            const uint32_t dst = s_gbase[(kv.x >> shift) & mask] + q;
            keys_out[dst] = kv.x;
            vals_out[dst] = kv.y;
            // Last pass of the capacity path (ranges != NULL; the keys are tile ids): the tile ranges come out of this scatter instead of
            // a launch of their own.  The tile's items are in sorted order here; a neighbour with the same key has the same digit and is the
            // global neighbour as well, so an item whose left (right) neighbour in the tile differs — or is missing — is the first (last)
            // of its key in this tile's run.  A key's items can sit in two tiles' runs: min / max through atomics on zeroed words, the
            // start kept as its complement (max of ~dst = min of dst); tile_depth_sort_kernel, the next launch, turns it back.
            if (ranges) {
                const bool first = q == 0u || s_item[q - 1].x != kv.x;
                const bool last = q + 1u == items || s_item[q + 1].x != kv.x;
                if (first) atomicMax(&ranges[kv.x].x, ~dst);
                if (last) atomicMax(&ranges[kv.x].y, dst + 1u);
            }
        }
    }
}

// The look-back head (ghist + tickets) has to be zero when a sort starts.  Callers whose previous kernel can clear
// radix_sort_head_words(n) words at the start of the scratch pass head_zeroed = true and save the memset launch.
size_t radix_sort_head_words(size_t n) { return rs_onesweep(n) ? RS_HEAD_WORDS : 0; }

// ---- inclusive scan of tiles[order[k]] (single launch, decoupled look-back; status word = flag | 30-bit value) -----------
constexpr int SC_IPT = 8, SC_TILE = RS_THREADS * SC_IPT;
size_t scan_scratch_words(size_t n) { return 64 + (n + SC_TILE - 1) / SC_TILE; }     // ticket (+pad) | status[tiles]; must be zero

__global__ void __launch_bounds__(RS_THREADS) scan_gather_kernel(const uint32_t* __restrict__ vals, const uint32_t* __restrict__ order, int packed,
                                                                 uint32_t* __restrict__ out, uint32_t n, uint32_t* __restrict__ scratch) {
    __shared__ uint32_t s_w[4];
    __shared__ uint32_t s_tile, s_excl;
    uint32_t* ticket = scratch;
    uint32_t* status = scratch + 64;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) s_tile = atomicAdd(ticket, 1u);
    __syncthreads();
    const uint32_t tile = s_tile;
    const uint32_t base = tile * SC_TILE + threadIdx.x * SC_IPT;
    uint32_t v[SC_IPT], sum = 0;
#pragma unroll
    for (int i = 0; i < SC_IPT; i++) {
        const uint32_t e = base + i;
        uint32_t t = 0u;
        if (e < n) {
            const uint32_t ov = order[e];
            if (packed) { t = ov >> PACK_ID_BITS; if (t == PACK_TILES_MAX) t = vals[ov & ((1u << PACK_ID_BITS) - 1u)]; }      // (the count rides in the sorted value)
            else t = vals[ov];
        }
        v[i] = t;
        sum += v[i];
    }
    uint32_t x = sum;                              // inclusive scan of the per-thread sums over the workgroup
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const uint32_t y = __shfl_up(x, o); if (lane >= o) x += y; }
    if (lane == 63) s_w[wave] = x;
    __syncthreads();
    uint32_t woff = 0;
    for (int w = 0; w < wave; w++) woff += s_w[w];
    const uint32_t agg = s_w[0] + s_w[1] + s_w[2] + s_w[3];
    if (wave == 0) {
        // wave 0 publishes the tile aggregate and looks back 64 tiles per round
        uint32_t excl = 0;
        if (tile == 0) {
            if (lane == 0) st_agent(status, agg | ST_PREFIX);
        } else {
            if (lane == 0) st_agent(status + tile, agg | ST_AGG);
            int p = (int)tile - 1;
            for (;;) {
                const int idx = p - lane;
                const uint32_t sw = idx >= 0 ? ld_agent(status + idx) : ST_PREFIX;
                const unsigned long long ready = __ballot((sw >> 30) != 0u);
                const unsigned long long pref = __ballot((sw & ST_PREFIX) != 0u);
                const int nready = ready == ~0ull ? 64 : __builtin_ctzll(~ready);       // leading run of published words
                const int firstp = pref ? __builtin_ctzll(pref) : 64;
                const int take = min(nready, firstp + 1);                                // up to and including the first prefix
                uint32_t c = lane < take ? (sw & ST_VALUE) : 0u;
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
                excl += c;
                if (firstp < nready) break;
                p -= take;
                if (take == 0) __builtin_amdgcn_s_sleep(1);
            }
            if (lane == 0) st_agent(status + tile, (excl + agg) | ST_PREFIX);
        }
        if (lane == 0) s_excl = excl;
    }
    __syncthreads();
    uint32_t run = s_excl + woff + x - sum;        // exclusive prefix of this thread's first item
#pragma unroll
    for (int i = 0; i < SC_IPT; i++) {
        run += v[i];
        if (base + i < n) out[base + i] = run;
    }
}

void launch_scan_gather(const uint32_t* vals, const uint32_t* order, int packed, uint32_t* out, size_t n, void* zeroed_scratch, hipStream_t s) {
    if (n == 0) return;
    hipLaunchKernelGGL(scan_gather_kernel, dim3((unsigned)((n + SC_TILE - 1) / SC_TILE)), dim3(RS_THREADS), 0, s, vals, order, packed, out, (uint32_t)n,
                       static_cast<uint32_t*>(zeroed_scratch));
}

// ---- per-tile depth sort ---------------------------------------------------------------------------------------------
// Small / medium frames (the BASELINE configs[1] regime): instead of depth-sorting all P surfels before emission (4 radix
// passes, ~16 us each because they are look-back-latency-bound, not byte-bound), instances are emitted in surfel-index order,
// grouped by the stable tile sort, and every tile then orders ITS OWN run by (float depth bits, surfel index) — the same total
// order the depth-presorted path produces — with a bitonic network in LDS: one launch, all tiles in parallel.
// Tiles with more than TS_CAP instances take a (slow, rare) chunked rank sort through global scratch.

// Bitonic network over 256*E keys held E per thread in registers (element index = e*256 + tid): partners 256 or more apart are
// another register of the same thread, partners closer than a wave come by lane shuffle, only distances 64 / 128 go through
// LDS.  Fully unrolled so that every register index is a compile-time constant.
template <int E>
__device__ __forceinline__ void tile_sort_regs(uint32_t n, uint32_t* __restrict__ pl, const uint32_t* __restrict__ depth_keys,
                                               unsigned long long* __restrict__ s, uint32_t tid) {
    unsigned long long v[E];
#pragma unroll
    for (int e = 0; e < E; e++) {
        const uint32_t idx = (uint32_t)e * RS_THREADS + tid;
        v[e] = ~0ull;
        if (idx < n) { const uint32_t id = pl[idx]; v[e] = ((unsigned long long)depth_keys[id] << 32) | id; }
    }
#pragma unroll
    for (uint32_t k = 2; k <= (uint32_t)RS_THREADS * E; k <<= 1) {
#pragma unroll
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            if (j >= (uint32_t)RS_THREADS) {
#pragma unroll
                for (int e = 0; e < E; e++) {
                    const int e2 = e | (int)(j / RS_THREADS);
                    if ((e & (int)(j / RS_THREADS)) == 0 && e2 < E) {
                        const bool up = ((((uint32_t)e * RS_THREADS + tid) & k) == 0u);
                        const unsigned long long a = v[e], b = v[e2];
                        if ((a > b) == up) { v[e] = b; v[e2] = a; }
                    }
                }
            } else {
                unsigned long long pv[E];
                if (j >= 64u) {
#pragma unroll
                    for (int e = 0; e < E; e++) s[e * RS_THREADS + tid] = v[e];
                    __syncthreads();
#pragma unroll
                    for (int e = 0; e < E; e++) pv[e] = s[e * RS_THREADS + (tid ^ j)];
                    __syncthreads();
                } else {
#pragma unroll
                    for (int e = 0; e < E; e++) pv[e] = __shfl_xor(v[e], (int)j, 64);
                }
#pragma unroll
                for (int e = 0; e < E; e++) {
                    const uint32_t idx = (uint32_t)e * RS_THREADS + tid;
                    const bool take_min = ((idx & j) == 0u) == ((idx & k) == 0u);
                    v[e] = ((pv[e] < v[e]) == take_min) ? pv[e] : v[e];
                }
            }
        }
    }
#pragma unroll
    for (int e = 0; e < E; e++) {
        const uint32_t idx = (uint32_t)e * RS_THREADS + tid;
        if (idx < n) pl[idx] = (uint32_t)v[e];
    }
}

// TS_CAP = largest run sorted by the LDS network (8 B of LDS per key: 2048 -> 16 KB -> 10 workgroups / CU for frames whose tiles
// are small; 4096 -> 32 KB otherwise).
template <int TS_CAP>
__global__ void __launch_bounds__(RS_THREADS) tile_depth_sort_kernel(uint2* __restrict__ ranges, uint32_t* __restrict__ point_list,
                                                                    const uint32_t* __restrict__ depth_keys, uint32_t* __restrict__ tmp_ids,
                                                                    uint32_t* __restrict__ tmp_keys, uint32_t* __restrict__ tmp_rank, int decode) {
    __shared__ unsigned long long s[TS_CAP];       // 32 KB: (depth bits << 32) | surfel index
    uint2 rg = ranges[blockIdx.x];
    if (decode) {      // ranges left by the sort's last pass (os_pass_fat_kernel): the start is stored complemented
        // every wave must HOLD the undecoded word before thread 0 overwrites it (a wave whose load landed after the store would complement
        // the start twice): readfirstlane consumes the loaded value in front of the barrier, the barrier orders all loads before the store
        rg.x = __builtin_amdgcn_readfirstlane(rg.x); rg.y = __builtin_amdgcn_readfirstlane(rg.y);
        __syncthreads();
        if (rg.y != 0u) {
            rg.x = ~rg.x;
            if (threadIdx.x == 0) ranges[blockIdx.x].x = rg.x;
        }
    }
    const uint32_t n = rg.y - rg.x;
    if (n < 2u) return;
    uint32_t* __restrict__ pl = point_list + rg.x;
    const uint32_t tid = threadIdx.x;
    // up to TS_CAP instances: E elements per thread, the network runs in registers (see tile_sort_regs)
    if (n <= (uint32_t)RS_THREADS) { tile_sort_regs<1>(n, pl, depth_keys, s, tid); return; }
    if (n <= 2u * RS_THREADS) { tile_sort_regs<2>(n, pl, depth_keys, s, tid); return; }
    if (n <= 4u * RS_THREADS) { tile_sort_regs<4>(n, pl, depth_keys, s, tid); return; }
    if (n <= 8u * RS_THREADS) { tile_sort_regs<8>(n, pl, depth_keys, s, tid); return; }
    if (TS_CAP >= 16 * RS_THREADS && n <= 16u * RS_THREADS) { tile_sort_regs<(TS_CAP >= 16 * RS_THREADS ? 16 : 8)>(n, pl, depth_keys, s, tid); return; }
    // fallback: rank of element i = number of elements with a smaller (depth, index) key; chunks of TS_CAP keys through LDS
    uint32_t* __restrict__ ids = tmp_ids + rg.x;
    uint32_t* __restrict__ keys = tmp_keys + rg.x;
    uint32_t* __restrict__ rank = tmp_rank + rg.x;
    for (uint32_t i = tid; i < n; i += RS_THREADS) { const uint32_t id = pl[i]; ids[i] = id; keys[i] = depth_keys[id]; rank[i] = 0u; }
    __syncthreads();
    for (uint32_t c0 = 0; c0 < n; c0 += (uint32_t)TS_CAP) {
        const uint32_t m = min((uint32_t)TS_CAP, n - c0);
        for (uint32_t i = tid; i < m; i += RS_THREADS) s[i] = ((unsigned long long)keys[c0 + i] << 32) | ids[c0 + i];
        __syncthreads();
        for (uint32_t i = tid; i < n; i += RS_THREADS) {
            const unsigned long long mine = ((unsigned long long)keys[i] << 32) | ids[i];
            uint32_t r = 0;
            for (uint32_t j = 0; j < m; j++) r += s[j] < mine ? 1u : 0u;
            rank[i] += r;
        }
        __syncthreads();
    }
    for (uint32_t i = tid; i < n; i += RS_THREADS) pl[rank[i]] = ids[i];
}

void launch_tile_depth_sort(int ntiles, int64_t R, uint2* ranges, uint32_t* point_list, const uint32_t* depth_keys, uint32_t* tmp_ids,
                            uint32_t* tmp_keys, uint32_t* tmp_rank, bool decode, hipStream_t st) {
    if (ntiles <= 0) return;
    if (R <= (int64_t)256 * ntiles)
        hipLaunchKernelGGL(tile_depth_sort_kernel<2048>, dim3(ntiles), dim3(RS_THREADS), 0, st, ranges, point_list, depth_keys, tmp_ids, tmp_keys,
                           tmp_rank, decode ? 1 : 0);
    else
        hipLaunchKernelGGL(tile_depth_sort_kernel<4096>, dim3(ntiles), dim3(RS_THREADS), 0, st, ranges, point_list, depth_keys, tmp_ids, tmp_keys,
                           tmp_rank, decode ? 1 : 0);
}

// Sorts on key bits [begin_bit, end_bit).  Buffers ping-pong a -> b -> a ...; returns 0 if the result is in (keys_a, vals_a),
// 1 if in (keys_b, vals_b); -1 if n is too large for the 30-bit look-back counters.
int radix_sort_pairs_u32(uint32_t* keys_a, uint32_t* vals_a, uint32_t* keys_b, uint32_t* vals_b, size_t n, int begin_bit, int end_bit,
                         void* scratch, hipStream_t s, bool head_zeroed) {
    if (n == 0 || end_bit <= begin_bit) return 0;      // an empty key field: the input order is the sorted order
    if (n >= ST_VALUE) return -1;
    const uint32_t nblocks = rs_nblocks(n);
    const int passes = radix_sort_passes(n, begin_bit, end_bit);
    if (passes > RS_MAX_PASSES) return -1;
    int cur = 0;
    if (use_rocprim(n, begin_bit, end_bit)) {
        size_t bytes = rocprim_sort_temp_bytes(n);
        if (rocprim::radix_sort_pairs(scratch, bytes, keys_a, keys_b, vals_a, vals_b, n, (unsigned)begin_bit, (unsigned)end_bit, s) != hipSuccess) return -1;
        return 1;
    }
    if (!rs_onesweep(n)) {
        uint32_t* hist = static_cast<uint32_t*>(scratch);
        uint32_t* total = hist + (size_t)RS_RADIX * nblocks;
        for (int p = 0; p < passes; p++) {
            const int bit = begin_bit + p * RS_BITS;
            const int nb = end_bit - bit < RS_BITS ? end_bit - bit : RS_BITS;
            const uint32_t mask = (1u << nb) - 1u;
            const uint32_t* kin = cur ? keys_b : keys_a; const uint32_t* vin = cur ? vals_b : vals_a;
            uint32_t* kout = cur ? keys_a : keys_b; uint32_t* vout = cur ? vals_a : vals_b;
            const uint32_t fb = ft_nblocks(n);
            hipLaunchKernelGGL(rs_hist_fat_kernel<FT_IPT>, dim3(fb), dim3(FTH_THREADS), 0, s, kin, (uint32_t)n, bit, mask, hist, fb);
            hipLaunchKernelGGL(rs_scan_kernel, dim3(RS_RADIX), dim3(RS_THREADS), 0, s, hist, fb, total);
            hipLaunchKernelGGL((os_pass_fat_kernel<false, FT_IPT>), dim3(fb), dim3(FT_THREADS), 0, s, kin, vin, kout, vout, (uint32_t)n, bit, mask, total, hist,
                               (uint32_t*)nullptr, (const uint32_t*)nullptr, 1, (uint2*)nullptr, fb);
            cur ^= 1;
        }
        return cur;
    }
    uint32_t* ghist = static_cast<uint32_t*>(scratch);
    uint32_t* ticket = ghist + RS_MAX_PASSES * RS_RADIX;
    uint32_t* status = ghist + RS_HEAD_WORDS;
    if (!head_zeroed) (void)hipMemsetAsync(scratch, 0, RS_HEAD_WORDS * sizeof(uint32_t), s);
    hipLaunchKernelGGL(os_hist_kernel, dim3(nblocks), dim3(RS_THREADS), 0, s, keys_a, (uint32_t)n, begin_bit, end_bit, passes, ghist, status, nblocks);
    for (int p = 0; p < passes; p++) {
        const int bit = begin_bit + p * RS_BITS;
        const int nb = end_bit - bit < RS_BITS ? end_bit - bit : RS_BITS;
        const uint32_t mask = (1u << nb) - 1u;
        const uint32_t* kin = cur ? keys_b : keys_a; const uint32_t* vin = cur ? vals_b : vals_a;
        uint32_t* kout = cur ? keys_a : keys_b; uint32_t* vout = cur ? vals_a : vals_b;
        hipLaunchKernelGGL((os_pass_fat_kernel<true, FT_IPT>), dim3(ft_nblocks(n)), dim3(FT_THREADS), 0, s, kin, vin, kout, vout, (uint32_t)n, bit, mask,
                           ghist + p * RS_RADIX, status + (size_t)p * nblocks * RS_RADIX, ticket + p, (const uint32_t*)nullptr, 1, (uint2*)nullptr, 0u);
        cur ^= 1;
    }
    return cur;
}

// ---- capacity binning (small / medium frames) ------------------------------------------------------------------------------
// The instance count R sizes the binning buffers, and round 2 made the host wait for it in the middle of the forward — a bubble
// of a wake-up and a few launches on the device every frame.  Here the buffers are sized from a CAPACITY (the largest count recent
// frames of this size produced, plus head room), everything is enqueued back to back, and the kernels read the real count from
// the device; the host only looks at R when the forward is fully enqueued (and falls back to the exact-size path in the rare
// frame that overflows).  With the buffers in place before the scan, three launches collapse into one:
//   bin_emit_kernel = inclusive scan of tiles_touched (surfel-index order, decoupled look-back) + emission of the (tile id,
//   surfel) pairs + the digit histograms of the tile sort + clearing of its look-back state.
// bin_emit_kernel — one thread per surfel (surfel-index order, workgroups of 256 like preprocess_fwd):
//   first instance slot = sum of the totals of the preprocess workgroups ahead (written there, 4 B per workgroup: no look-back
//   chain, no ticket) + an in-block scan;  emission of the (tile id, surfel) pairs;  digit histograms of the tile sort in LDS,
//   flushed with one atomic per (workgroup, non-empty bin) into bins padded to a cache line each (same-line atomics serialise:
//   1172 workgroups x 272 bins on 17 lines would cost ~20 us, on 272 lines ~1 us);  clearing of the sort's look-back state.
constexpr int BE_HSTRIDE = 32;      // words between histogram bins of the capacity path (128 B)
size_t bin_emit_head_words() { return (size_t)RS_MAX_PASSES * RS_RADIX * BE_HSTRIDE + 64; }      // padded ghist | tickets: must be zero

constexpr int BE_THREADS = 1024;      // 16 waves per workgroup: a quarter of the (workgroup, bin) histogram atomics of 256-thread workgroups
__global__ void __launch_bounds__(BE_THREADS) bin_emit_kernel(int P, const uint32_t* __restrict__ tiles_touched, const uint32_t* __restrict__ rects,
                                                              const uint32_t* __restrict__ block_totals, uint32_t nblock_totals, float* __restrict__ rec,
                                                              uint32_t* __restrict__ keys, uint32_t* __restrict__ vals, int gx, uint32_t cap,
                                                              uint32_t* __restrict__ ghist, uint32_t* __restrict__ sort_status,
                                                              uint32_t sort_status_words, int passes, int end_bit, uint32_t* __restrict__ n_out,
                                                              uint32_t* __restrict__ n_host) {
    constexpr int NW = BE_THREADS / 64;
    __shared__ uint32_t s_w[NW], s_p[NW];
    __shared__ uint32_t s_h[2][RS_RADIX];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t k = blockIdx.x * BE_THREADS + threadIdx.x;
    if (threadIdx.x < 2 * RS_RADIX) s_h[threadIdx.x >> 8][threadIdx.x & (RS_RADIX - 1)] = 0u;
    for (uint32_t w = k; w < sort_status_words; w += gridDim.x * BE_THREADS) sort_status[w] = 0u;
    // Every addition of the scan SATURATES at 0xffffffff: a frame with >= 2^32 tile instances (far beyond any capacity: <= 2^20) must
    // come out as "overflowed", not as a small wrapped total that passes the host's R <= cap check with truncated lists.
    auto sat = [](uint32_t a, uint32_t b) { const uint32_t c = a + b; return c < a ? 0xffffffffu : c; };
    uint32_t pre = 0;                              // instances of the surfels ahead of this workgroup (preprocess workgroups of 256)
    const uint32_t ahead = min(nblock_totals, blockIdx.x * (uint32_t)(BE_THREADS / 256));
    for (uint32_t j = threadIdx.x; j < ahead; j += BE_THREADS) pre = sat(pre, block_totals[j]);
    const uint32_t n = k < (uint32_t)P ? tiles_touched[k] : 0u;
    uint32_t x = n;                                // inclusive scan over the workgroup
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const uint32_t y = __shfl_up(x, o); if (lane >= o) x = sat(x, y); }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) pre = sat(pre, __shfl_xor(pre, o));
    if (lane == 63) s_w[wave] = x;
    if (lane == 0) s_p[wave] = pre;
    __syncthreads();
    uint32_t off = x - n;                          // (x saturated: off is an under-estimate >= 2^32 - 1 - n, still far above any capacity)
#pragma unroll
    for (int w = 0; w < NW; w++) off = sat(off, sat(s_p[w], w < wave ? s_w[w] : 0u));
    if (k == (uint32_t)P - 1u) {
        n_out[0] = sat(off, n);      // the instance total, for the kernels that follow ...
        if (n_host) n_host[0] = sat(off, n);      // ... and for the host (mapped pinned word: no copy kernel in the stream)
    }
    const uint32_t mask0 = (1u << min(RS_BITS, end_bit)) - 1u;
    const uint32_t mask1 = passes > 1 ? (1u << min(RS_BITS, end_bit - RS_BITS)) - 1u : 0u;
    if (n) {
        const uint32_t rectbits = rects[k];
        rec[(size_t)k * REC_F + 18] = __uint_as_float(off);      // the record's first-instance slot
        const int x0 = rectbits & 1023, y0 = (rectbits >> 10) & 1023, w = rectbits >> 20;
        int xx = 0, yy = 0;
        for (uint32_t t = 0; t < n; t++) {
            const uint32_t key = (uint32_t)((y0 + yy) * gx + (x0 + xx));
            if (off < cap && t < cap - off) {      // (an overflowing frame is redone by the caller; what is sorted here stays consistent and in bounds)
                keys[off + t] = key; vals[off + t] = k;
                atomicAdd(&s_h[0][key & mask0], 1u);
                if (passes > 1) atomicAdd(&s_h[1][(key >> RS_BITS) & mask1], 1u);
            }
            if (++xx == w) { xx = 0; ++yy; }
        }
    }
    __syncthreads();
    if ((int)threadIdx.x < passes * RS_RADIX) {
        const uint32_t c = s_h[threadIdx.x >> 8][threadIdx.x & (RS_RADIX - 1)];
        if (c) atomicAdd(&ghist[(size_t)threadIdx.x * BE_HSTRIDE], c);
    }
}

// tile sort of the capacity path: the look-back passes alone (histograms and state come from bin_emit_kernel); at most two passes
int radix_sort_pairs_u32_devn(uint32_t* keys_a, uint32_t* vals_a, uint32_t* keys_b, uint32_t* vals_b, size_t cap, int end_bit, const uint32_t* n_dev,
                              void* scratch, uint2* ranges, hipStream_t s) {
    // scratch layout of the capacity path (u32 words): ghist[RS_MAX_PASSES][256] padded to BE_HSTRIDE | ticket (+pad to 64) | status
    const uint32_t nblocks = rs_nblocks(cap);
    const int passes = radix_sort_passes(cap, 0, end_bit);
    uint32_t* ghist = static_cast<uint32_t*>(scratch);
    uint32_t* ticket = ghist + (size_t)RS_MAX_PASSES * RS_RADIX * BE_HSTRIDE;
    uint32_t* status = ghist + bin_emit_head_words();
    int cur = 0;
    for (int p = 0; p < passes; p++) {
        const int bit = p * RS_BITS;
        const int nb = end_bit - bit < RS_BITS ? end_bit - bit : RS_BITS;
        const uint32_t mask = (1u << nb) - 1u;
        const uint32_t* kin = cur ? keys_b : keys_a; const uint32_t* vin = cur ? vals_b : vals_a;
        uint32_t* kout = cur ? keys_a : keys_b; uint32_t* vout = cur ? vals_a : vals_b;
        hipLaunchKernelGGL((os_pass_fat_kernel<true, FT_IPT>), dim3(ft_nblocks(cap)), dim3(FT_THREADS), 0, s, kin, vin, kout, vout, (uint32_t)cap, bit, mask,
                           ghist + (size_t)p * RS_RADIX * BE_HSTRIDE, status + (size_t)p * nblocks * RS_RADIX, ticket + p, n_dev, BE_HSTRIDE,
                           p == passes - 1 ? ranges : (uint2*)nullptr, 0u);      // (the last pass leaves the tile ranges behind, start complemented)
        cur ^= 1;
    }
    return cur;
}
bool capacity_binning_ok(size_t cap, int end_bit) { return rs_onesweep(cap) && radix_sort_passes(cap, 0, end_bit) <= 2 && g_large_sort_impl != 3; }
size_t bin_emit_sort_status_words(size_t cap, int end_bit) { return (size_t)radix_sort_passes(cap, 0, end_bit) * rs_nblocks(cap) * RS_RADIX; }
size_t capacity_sort_scratch_bytes(size_t cap, int end_bit) { return (bin_emit_head_words() + bin_emit_sort_status_words(cap, end_bit)) * sizeof(uint32_t) + 256; }

void launch_bin_emit(int P, const uint32_t* tiles_touched, const uint32_t* rects, const uint32_t* block_totals, float* rec, uint32_t* keys, uint32_t* vals,
                     int gx, size_t cap, void* sort_scratch, int end_bit, uint32_t* n_out, uint32_t* n_host, hipStream_t s) {
    uint32_t* ghist = static_cast<uint32_t*>(sort_scratch);
    hipLaunchKernelGGL(bin_emit_kernel, dim3((unsigned)((P + BE_THREADS - 1) / BE_THREADS)), dim3(BE_THREADS), 0, s, P, tiles_touched, rects, block_totals,
                       (uint32_t)((P + 255) / 256), rec, keys, vals, gx, (uint32_t)cap, ghist, ghist + bin_emit_head_words(), (uint32_t)bin_emit_sort_status_words(cap, end_bit),
                       radix_sort_passes(cap, 0, end_bit), end_bit, n_out, n_host);
}

}  // namespace surfel
