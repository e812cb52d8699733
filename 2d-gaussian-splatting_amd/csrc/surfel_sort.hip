// surfel_sort.hip — stable LSD radix sort of (u32 key, u32 value) pairs for the binning stages, gfx950.
//
// Why not rocPRIM here: the two sorts on the hot path are small and oddly shaped — P surfels by 32 depth bits
// and R instances by the 12-16 tile-id bits — and rocPRIM's generic dispatch costs ~0.14 ms each at the
// BASELINE configs[1] size (300 k surfels / 0.6 M instances).  This version is 3 launches per pass (8- or 11-bit digits):
//   histogram  : per-block digit counts  -> hist[digit][block]
//   scan       : one workgroup per digit, exclusive scan over blocks, digit total -> total[digit]
//   scatter    : wave64 match-by-ballot ranks (no atomics, order-preserving => stable), coalesced-run stores
// Traffic per pass: 4 B (hist) + 8 B read + 8 B written per element; all integer, HBM-streaming work.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "surfel_kernels.h"

namespace surfel {

constexpr int RS_THREADS = 256;
constexpr int RS_IPT = 8;                       // items per thread
constexpr int RS_TILE = RS_THREADS * RS_IPT;    // 2048 items per workgroup
constexpr int RS_FUSED_MAX_BLOCKS = 1024;       // up to here the scatter kernel scans the block histograms itself

// digit histogram of one 2048-item block -> hist[digit][block]
template <int BITS>
__global__ void __launch_bounds__(RS_THREADS) rs_hist_kernel(const uint32_t* __restrict__ keys, uint32_t n, int shift, uint32_t mask,
                                                             uint32_t* __restrict__ hist, uint32_t nblocks) {
    constexpr int RADIX = 1 << BITS;
    __shared__ uint32_t s_h[RADIX];
    for (int d = threadIdx.x; d < RADIX; d += RS_THREADS) s_h[d] = 0;
    __syncthreads();
    const uint32_t base = blockIdx.x * RS_TILE;
#pragma unroll
    for (int it = 0; it < RS_IPT; it++) {
        const uint32_t e = base + it * RS_THREADS + threadIdx.x;
        if (e < n) atomicAdd(&s_h[(keys[e] >> shift) & mask], 1u);
    }
    __syncthreads();
    for (int d = threadIdx.x; d < RADIX; d += RS_THREADS) hist[(size_t)d * nblocks + blockIdx.x] = s_h[d];
}

// one workgroup per digit: in-place exclusive scan of hist[digit][0..nblocks), total[digit] = sum  (large-n path)
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

// Stable scatter of one block.  FUSED: the block derives its global digit offsets straight from the raw block
// histograms (sum of the blocks before it + full-row totals) — no separate scan launch; used for small n.
template <int BITS, bool FUSED>
__global__ void __launch_bounds__(RS_THREADS) rs_scatter_kernel(const uint32_t* __restrict__ keys, const uint32_t* __restrict__ vals,
                                                                uint32_t* __restrict__ keys_out, uint32_t* __restrict__ vals_out, uint32_t n,
                                                                int shift, uint32_t mask, const uint32_t* __restrict__ hist,
                                                                const uint32_t* __restrict__ total, uint32_t nblocks) {
    constexpr int RADIX = 1 << BITS;
    constexpr int DPT = RADIX / RS_THREADS;       // digits owned per thread in the offset phases (contiguous)
    __shared__ uint32_t s_cnt[4][RADIX];          // per-wave digit counts -> then per-wave output offsets
    __shared__ uint32_t s_base[RADIX];
    __shared__ uint32_t s_w[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // global base of every digit = exclusive scan of the digit totals + this block's offset inside the digit
    {
        uint32_t tot[DPT], mine[DPT], run = 0;
#pragma unroll
        for (int q = 0; q < DPT; q++) {
            const uint32_t d = threadIdx.x * DPT + q;
            if (FUSED) {
                const uint32_t* row = hist + (size_t)d * nblocks;
                uint32_t before = 0, all = 0;
                for (uint32_t b = 0; b < nblocks; b++) { const uint32_t c = row[b]; all += c; before += (b < blockIdx.x) ? c : 0u; }
                tot[q] = all; mine[q] = before;
            } else {
                tot[q] = total[d]; mine[q] = hist[(size_t)d * nblocks + blockIdx.x];
            }
            run += tot[q];
#pragma unroll
            for (int w = 0; w < 4; w++) s_cnt[w][d] = 0;
        }
        uint32_t x = run;                          // inclusive scan of per-thread totals across the block
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const uint32_t y = __shfl_up(x, o); if (lane >= o) x += y; }
        if (lane == 63) s_w[wave] = x;
        __syncthreads();
        uint32_t off = x - run;
        for (int w = 0; w < wave; w++) off += s_w[w];
#pragma unroll
        for (int q = 0; q < DPT; q++) { s_base[threadIdx.x * DPT + q] = off + mine[q]; off += tot[q]; }
    }
    // phase A: wave w owns items [w*512, w*512+512) of the tile, 8 sweeps of 64 consecutive items
    const uint32_t wbase = blockIdx.x * RS_TILE + wave * (64 * RS_IPT);
    uint32_t k[RS_IPT], v[RS_IPT], rank[RS_IPT];
    const unsigned long long lt = (1ull << lane) - 1ull;
#pragma unroll
    for (int it = 0; it < RS_IPT; it++) {
        const uint32_t e = wbase + it * 64 + lane;
        const bool valid = e < n;
        k[it] = valid ? keys[e] : 0xffffffffu;
        v[it] = valid ? vals[e] : 0u;
        const uint32_t d = (k[it] >> shift) & mask;
        unsigned long long mm = __ballot(valid);
#pragma unroll
        for (int b = 0; b < BITS; b++) {
            const bool bit = (d >> b) & 1u;
            const unsigned long long bal = __ballot(bit);
            mm &= bit ? bal : ~bal;
        }
        const uint32_t prev = s_cnt[wave][d];                 // running count of this digit in this wave (broadcast read)
        const uint32_t before = (uint32_t)__popcll(mm & lt);
        rank[it] = prev + before;
        if (valid && before == 0) s_cnt[wave][d] = prev + (uint32_t)__popcll(mm);   // group leader bumps the counter
    }
    __syncthreads();
    // phase B: per-wave output offsets for every digit
#pragma unroll
    for (int q = 0; q < DPT; q++) {
        const uint32_t d = threadIdx.x * DPT + q;
        uint32_t off = s_base[d];
#pragma unroll
        for (int w = 0; w < 4; w++) { const uint32_t c = s_cnt[w][d]; s_cnt[w][d] = off; off += c; }
    }
    __syncthreads();
    // phase C: scatter
#pragma unroll
    for (int it = 0; it < RS_IPT; it++) {
        const uint32_t e = wbase + it * 64 + lane;
        if (e < n) {
            const uint32_t d = (k[it] >> shift) & mask;
            const uint32_t pos = s_cnt[wave][d] + rank[it];
            keys_out[pos] = k[it];
            vals_out[pos] = v[it];
        }
    }
}

// Digit width: small inputs are launch-bound (every launch costs ~4.5 us of dispatch floor), so they take 11-bit digits
// = fewer passes; large inputs take 8-bit digits (smaller per-block histograms, longer same-digit store runs).
// (A scatter that scans the block histograms itself — the FUSED template path — was measured NOT to pay: its
// O(radix x nblocks) prologue per block costs more than the 4.8 us scan launch it saves.)
static bool rs_small(size_t n) { return (n + RS_TILE - 1) / RS_TILE <= RS_FUSED_MAX_BLOCKS; }
static int rs_bits(size_t n) { return rs_small(n) ? 11 : 8; }

size_t radix_sort_scratch_bytes(size_t n) {
    const size_t nblocks = (n + RS_TILE - 1) / RS_TILE;
    const size_t radix = (size_t)1 << rs_bits(n);
    return (radix * nblocks + radix) * sizeof(uint32_t) + 256;
}

int radix_sort_passes(size_t n, int begin_bit, int end_bit) {
    const int bits = rs_bits(n);
    return (end_bit - begin_bit + bits - 1) / bits;
}

// Sorts on key bits [begin_bit, end_bit).  Buffers ping-pong a -> b -> a ...; returns 0 if the result is in (keys_a, vals_a),
// 1 if in (keys_b, vals_b).
int radix_sort_pairs_u32(uint32_t* keys_a, uint32_t* vals_a, uint32_t* keys_b, uint32_t* vals_b, size_t n, int begin_bit, int end_bit,
                         void* scratch, hipStream_t s) {
    if (n == 0) return 0;
    const uint32_t nblocks = (uint32_t)((n + RS_TILE - 1) / RS_TILE);
    const bool small = rs_small(n);
    const int bits = rs_bits(n);
    const int passes = radix_sort_passes(n, begin_bit, end_bit);
    const int per = (end_bit - begin_bit + passes - 1) / passes;          // spread the bits evenly over the passes
    uint32_t* hist = static_cast<uint32_t*>(scratch);
    uint32_t* total = hist + ((size_t)1 << bits) * nblocks;
    int cur = 0;
    for (int bit = begin_bit; bit < end_bit; bit += per) {
        const int nb = end_bit - bit < per ? end_bit - bit : per;
        const uint32_t mask = (1u << nb) - 1u;
        const uint32_t* kin = cur ? keys_b : keys_a; const uint32_t* vin = cur ? vals_b : vals_a;
        uint32_t* kout = cur ? keys_a : keys_b; uint32_t* vout = cur ? vals_a : vals_b;
        if (small) {
            hipLaunchKernelGGL(rs_hist_kernel<11>, dim3(nblocks), dim3(RS_THREADS), 0, s, kin, (uint32_t)n, bit, mask, hist, nblocks);
            hipLaunchKernelGGL(rs_scan_kernel, dim3(2048), dim3(RS_THREADS), 0, s, hist, nblocks, total);
            hipLaunchKernelGGL((rs_scatter_kernel<11, false>), dim3(nblocks), dim3(RS_THREADS), 0, s, kin, vin, kout, vout, (uint32_t)n, bit, mask,
                               hist, total, nblocks);
        } else {
            hipLaunchKernelGGL(rs_hist_kernel<8>, dim3(nblocks), dim3(RS_THREADS), 0, s, kin, (uint32_t)n, bit, mask, hist, nblocks);
            hipLaunchKernelGGL(rs_scan_kernel, dim3(256), dim3(RS_THREADS), 0, s, hist, nblocks, total);
            hipLaunchKernelGGL((rs_scatter_kernel<8, false>), dim3(nblocks), dim3(RS_THREADS), 0, s, kin, vin, kout, vout, (uint32_t)n, bit, mask,
                               hist, total, nblocks);
        }
        cur ^= 1;
    }
    return cur;
}

}  // namespace surfel
