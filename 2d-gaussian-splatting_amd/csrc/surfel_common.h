// surfel_common.h — shared definitions for the gfx950 surfel rasterizer kernels.
// Written for CDNA4 only: wave64, 16x16-pixel tiles = 4 waves, each wave owning an 8x8 pixel quad.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace surfel {

constexpr int TILE = 16;            // tile edge in pixels (fixed by the reference's binning semantics)
constexpr int BLOCK = TILE * TILE;  // 256 threads = 4 waves
constexpr int WAVE = 64;
constexpr int R_SLOTS = 64;         // partial instance totals (summed on the host)
constexpr int PACK_ID_BITS = 22;    // frames of < 2^22 surfels: the depth sort's value = surfel id | min(tiles_touched, PACK_TILES_MAX) << 22
constexpr uint32_t PACK_TILES_MAX = 1023u;      // (this many or more: the scan looks the count up)

// rasterizer constants (oracle/surfel_oracle.c holds the same list with provenance)
constexpr float NEAR_N = 0.2f;
constexpr float FAR_N = 100.0f;
constexpr float FILTER_SIZE = 0.707106f;
constexpr float FILTER_INV_SQUARE = 2.0f;
constexpr float CUTOFF = 3.0f;
constexpr float ALPHA_MIN = 1.0f / 255.0f;
constexpr float ALPHA_MAX = 0.99f;
constexpr float T_EPS = 0.0001f;

// Per-surfel packed record, 112 B = 7 x float4, written by preprocess, gathered by the blend kernels
// (q0-q4 are staged in LDS for the blend loop; q5-q6 feed the staging-time culling only).
//   q0 = Tu.x Tu.y Tu.z Tv.x
//   q1 = Tv.y Tv.z Tw.x Tw.y
//   q2 = Tw.z xy.x xy.y opacity
//   q3 = n.x  n.y  n.z  r
//   q4 = g    b    inst_base(u32 bits)  rect(u32 bits: x0 | y0<<10 | w<<20)
//   q5 = ecx  ecy  Sxx  Sxy     conservative footprint of {alpha >= 1/255} (cull only, never changes results):
//   q6 = Syy  r2^2 det  -       ellipse {d^T S^-1 d <= 1} about (ecx,ecy)  U  disc of radius r2 about xy
#ifndef SURFEL_REC_F
#define SURFEL_REC_F 28
#endif
constexpr int REC_F = SURFEL_REC_F;      // floats per record: 28 (packed) or 32 (one 128-B line per record; measured: profiles/r03_record_stride.md)
constexpr float FOOT_UNBOUNDED = 1.0e30f;   // Sxx >= this: the footprint is the whole image
constexpr int REC_Q = REC_F / 4;
// Per-(tile,surfel) gradient record written by blend-backward, summed by preprocess-backward:
//   [0..8] dL/dT (Tu,Tv,Tw)  [9..10] dL/dxy (low-pass branch)  [11..13] dL/dnormal  [14] dL/dopacity
//   [15..17] dL/drgb  [18..19] pad
constexpr int GREC_F = 20;

// The blend kernels use the hardware's approximate reciprocal (v_rcp_f32, 1 ulp) and exp2-based exponential (v_exp_f32);
// -DSURFEL_IEEE_MATH (python build.py --ieee -> lib/libsurfel_hip_ieee.so, diagnostics only) swaps in correctly rounded division
// and libm-grade expf so that what the fast forms cost in parity can be measured (profiles/r02_fast_intrinsics_cost.md).
#ifdef SURFEL_IEEE_MATH
#define SURFEL_RCP(x) (1.0f / (x))
#define SURFEL_GAUSS(rho) expf(-0.5f * (rho))
#else
#define SURFEL_RCP(x) __builtin_amdgcn_rcpf(x)
// exp(-rho / 2) = 2^(rho * -log2(e) / 2): ONE multiply in front of v_exp_f32 (round 6; -0.5f * rho through __expf took two)
#define SURFEL_GAUSS(rho) __builtin_amdgcn_exp2f(-0.72134752044448170368f * (rho))
#endif

// ---- ray-splat intersection of one (pixel, surfel) pair --------------------------------------------
// ONE definition for blend_fwd and every blend_bwd walk: the backward re-decides, pair by pair, what the forward composited
// (alpha >= 1/255, depth >= near, p2 != 0), so both sides must round identically.  Every fused multiply-add is spelled out and
// contraction is switched off inside these functions, whatever -ffp-contract the including file is compiled with
// (tests/test_gpu_parity.py::test_forward_and_backward_composite_the_same_pairs counts both sides).
#define SURFEL_FMA(a, b, c) __builtin_fmaf((a), (b), (c))
struct Hit {
    float kx, ky, kz, lx, ly, lz, sx, sy, ip, dx, dy, depth, G, alpha, Twx, Twy, opa;
    bool use3d;
};
// k = px*Tw - Tu, l = py*Tw - Tv and the offsets to the projected centre (the low-pass branch)
__device__ __forceinline__ void pair_planes(float pxf, float pyf, const float4 q0, const float4 q1, const float4 q2, Hit& h) {
#pragma clang fp contract(off)
    const float Twx = q1.z, Twy = q1.w, Twz = q2.x;
    h.kx = SURFEL_FMA(pxf, Twx, -q0.x); h.ky = SURFEL_FMA(pxf, Twy, -q0.y); h.kz = SURFEL_FMA(pxf, Twz, -q0.z);
    h.lx = SURFEL_FMA(pyf, Twx, -q0.w); h.ly = SURFEL_FMA(pyf, Twy, -q1.x); h.lz = SURFEL_FMA(pyf, Twz, -q1.y);
    h.dx = q2.y - pxf; h.dy = q2.z - pyf;
}
// from the planes: intersection (s), depth, alpha; returns the forward's compositing tests (p2 != 0, depth >= near, alpha >= 1/255)
__device__ __forceinline__ bool pair_intersect(const float Twx, const float Twy, const float Twz, const float opa, Hit& h) {
#pragma clang fp contract(off)
    h.Twx = Twx; h.Twy = Twy; h.opa = opa;
    const float p0 = SURFEL_FMA(h.ky, h.lz, -(h.kz * h.ly)), p1 = SURFEL_FMA(h.kz, h.lx, -(h.kx * h.lz)), p2 = SURFEL_FMA(h.kx, h.ly, -(h.ky * h.lx));
    h.ip = SURFEL_RCP(p2);
    h.sx = p0 * h.ip; h.sy = p1 * h.ip;
    const float rho3d = SURFEL_FMA(h.sx, h.sx, h.sy * h.sy);
    const float rho2d = FILTER_INV_SQUARE * SURFEL_FMA(h.dx, h.dx, h.dy * h.dy);
    h.use3d = rho3d <= rho2d;
    const float rho = fminf(rho3d, rho2d);
    h.depth = h.use3d ? SURFEL_FMA(h.sx, Twx, SURFEL_FMA(h.sy, Twy, Twz)) : Twz;      // (round 6: two fused multiply-adds; sx Twx + sy Twy, + Twz took three operations)
    h.G = SURFEL_GAUSS(rho);
    h.alpha = fminf(ALPHA_MAX, opa * h.G);
    return (p2 != 0.f) & (h.depth >= NEAR_N) & (h.alpha >= ALPHA_MIN);
}
__device__ __forceinline__ bool pair_hit(float pxf, float pyf, const float4 q0, const float4 q1, const float4 q2, Hit& h) {
    pair_planes(pxf, pyf, q0, q1, q2, h);
    return pair_intersect(q1.z, q1.w, q2.x, q2.w, h);
}

// A lazily counted capacity-path frame whose instance total exceeded its capacity (include/surfel_hip.h: SURFEL_OPT_LAZY_COUNT): its
// lists are truncated and its records' first-instance slots run past the gradient-record allocation.  The caller redoes the frame
// once it has collected the count; until then every backward kernel it may already have enqueued must touch nothing.
__device__ __forceinline__ bool frame_overflowed(const uint32_t* __restrict__ n_dev, uint32_t n_cap) {
#ifdef SURFEL_NO_OVERFLOW_GUARD      // diagnostic build: shows that tests/test_gpu_guard.py faults without the check (python build.py --variant noguard -DSURFEL_NO_OVERFLOW_GUARD)
    return false;
#else
    return n_dev != nullptr && n_dev[0] > n_cap;
#endif
}

// ---- the tile stream (round 5) ---------------------------------------------------------------------------------------------------
// blend_bwd used to stage a batch as blend_fwd does: surfel ids of the list -> 112-B record gather (two DEPENDENT global round trips,
// 64 cache lines per load instruction, every line fetched up to seven times) -> exact footprint test of every instance against the
// tile's 16 sub-tiles (~150 instructions per staged instance) — 28 % of a scan-walk wave's life on a trained frame
// (profiles/r04_wg_trace.md section 4), and the backward has no LDS left for the forward's remedy (a second record buffer filled by DMA).
// But the forward HAS all of it in LDS at the moment it walks a batch: the whole records and the footprint bits.  It now leaves them
// behind in list order — 80 B (q0-q4) + 4 B (the 16 sub-tile bits, in the backward's row order) per staged list position, written
// once, coalesced, by the wave that fetched them — and the backward's staging becomes ONE contiguous, dependency-free read per batch,
// prefetched a batch ahead.  The blend kernels run at < 0.1 of the HBM roofline and are bound by issue and latency: this trades
// bytes (164 B per staged instance, about what the gather fetched) for both.  Only positions the forward walked have entries — exactly
// the positions (<= a tile's deepest contributor) the backward stages.  Frames above 2^24 instances (C5: 11 GB of stream for 4 % staged
// positions) and the batch-synchronous forward kernel write none: the two words the forward leaves in the image buffer say where the
// stream is, 0 = nowhere, and the backward gathers as before — same bits either way (tests/test_gpu_parity.py::test_tile_stream_*).
constexpr int STRM_Q = 5;                          // float4s per stream record (q0-q4 of the 112-B surfel record)
constexpr long long STRM_MAX_R = 1ll << 24;        // largest binning capacity that gets a stream
// sub-tile bits in `sub` order (4 * by + bx, blend_fwd) -> row order (bit 4 w + r <-> DPP row r of wave w, blend_bwd): swap index bits 1, 2
__device__ __forceinline__ unsigned subtile_bits_to_rows(unsigned m) {
    return (m & 0xC3C3u) | ((m & 0x0C0Cu) << 2) | ((m & 0x3030u) >> 2);
}

// Which blend-backward walk a frame takes under "bwd_variant" = auto — a rule on the frame's own totals, so that the bits of a frame
// follow from the frame alone (the walks differ in summation order).  The scan walk (surfel_backward_scan.hip) wins where a surfel's
// footprint spans many tiles — trained frames: 9 instances per emitting surfel, -9 % against rows; C2H: 10.8, -3 ... -6 % — and on large
// frames (2^21 <= R < 2^26: C4, garden -7 %; C5, 1.3e8 instances of which 4 % are staged: no gain); small random footprints (C2: 2.2
// instances per surfel) stay with the per-row walk (scan +11 % there).  Evaluated ONCE per frame, by the first wave of blend_fwd, and
// left in a word of the image buffer's head: the backward's two grids each read that word (round 5: every workgroup of both grids used
// to sum the 128 partial counters itself, in front of its first load).
constexpr unsigned SCAN_MIN_INST_PER_SURFEL = 6;
constexpr uint32_t WALK_ROWS = 1u, WALK_SCAN = 2u;
__device__ __forceinline__ uint32_t frame_walk(const uint32_t* __restrict__ totals /* [2 * R_SLOTS]: tile instances | emitting surfels */) {
    const int lane = threadIdx.x & 63;
    uint32_t r = totals[lane], v = totals[R_SLOTS + lane];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { r += __shfl_xor(r, o); v += __shfl_xor(v, o); }
    const unsigned long long R = r;
    return (R < (1ull << 26) && (R >= (1ull << 21) || R >= (unsigned long long)SCAN_MIN_INST_PER_SURFEL * v)) ? WALK_SCAN : WALK_ROWS;
}

struct Rect { int x0, y0, x1, y1; };

__device__ __forceinline__ Rect tile_rect(float px, float py, int r, int gx, int gy) {
    Rect rc;
    rc.x0 = min(gx, max(0, (int)((px - r) / TILE)));
    rc.y0 = min(gy, max(0, (int)((py - r) / TILE)));
    rc.x1 = min(gx, max(0, (int)((px + r + TILE - 1) / TILE)));
    rc.y1 = min(gy, max(0, (int)((py + r + TILE - 1) / TILE)));
    return rc;
}

// Tile index for a workgroup: block b is dispatched to XCD b % 8 (observed, speed only), so give each
// XCD a contiguous run of tiles — neighbouring tiles share surfel records, which then hit in that
// XCD's private L2.  Bijective for any tile count.
__device__ __forceinline__ int xcd_tile(int b, int n) {
    const int q = n >> 3, r = n & 7;
    const int xcd = b & 7, i = b >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + i;
}

// The tile a blend workgroup takes: the frame's longest-first map where tile_order_kernel wrote one (surfel_preprocess.hip), the
// XCD-contiguous order otherwise; -1: none (the grid covers the map's length).
__device__ __forceinline__ int block_tile(const int* __restrict__ map, const uint32_t* __restrict__ map_flag, int b, int n) {
    if (map_flag[0] != 0u) return map[b];
    return b < n ? xcd_tile(b, n) : -1;
}

// ---- footprint culling ---------------------------------------------------------------------------
// The pixels a surfel can touch with alpha >= 1/255 lie inside  E U D :  E = the projected sqrt(rmax)-sigma ellipse
// {d = p - e : d^T S^-1 d <= 1} (rho3d <= rmax), D = the low-pass disc (rho2d <= rmax), both inflated for fp32
// (preprocess_fwd).  Culling is done ONCE per (tile, instance) by the staging thread, i.e. at 1/64 of the price of
// a blend visit, so it can afford the exact geometry: for a horizontal strip of pixel rows the x-interval of
// (E U D) /\ strip is closed-form (the extreme points of an ellipse inside a strip are its global extreme points
// clamped to the strip), and a block of pixel columns is hit iff it meets that interval.
// v_sqrt_f32 (1 ulp) instead of the correctly rounded sqrtf (a dozen instructions with its refinement and denormal scaling; 13 of
// them per staged instance were half of the staging arithmetic): the footprint is conservative by construction — the ellipse is
// inflated by 0.2 % + 0.3 px in preprocess, every interval by 1e-3 px here — so an ulp of the half-width changes no culling decision
// that matters (tests/test_gpu_parity.py::test_culling_is_exact holds the images to the uncullled bits).
__device__ __forceinline__ float foot_sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
struct Foot {
    float ecx, ecy, syy, det, k, isyy, hy, yR;   // ellipse
    float dcx, dcy, r2sq;                        // disc
    bool unbounded;
};
__device__ __forceinline__ Foot make_foot(const float4 q2, const float4 q5, const float4 q6) {
    Foot f;
    f.ecx = q5.x; f.ecy = q5.y; f.syy = q6.x; f.det = q6.z;
    f.unbounded = q5.z >= FOOT_UNBOUNDED;
    f.isyy = __builtin_amdgcn_rcpf(q6.x);
    f.k = q5.w * f.isyy;                         // dx/dy of the ellipse's centre line
    f.hy = foot_sqrt(q6.x);
    f.yR = q5.w * __builtin_amdgcn_rsqf(q5.z);   // y of the rightmost point (leftmost: -yR)
    f.dcx = q2.y; f.dcy = q2.z; f.r2sq = q6.y;
    return f;
}
// x-interval [xmin, xmax] of (E U D) within the closed strip y in [Y0, Y1] (empty: xmin > xmax)
__device__ __forceinline__ void foot_strip(const Foot& f, float Y0, float Y1, float& xmin, float& xmax) {
    xmin = 3.0e38f; xmax = -3.0e38f;
    const float a0 = fmaxf(Y0 - f.ecy, -f.hy), a1 = fminf(Y1 - f.ecy, f.hy);
    if (a0 <= a1) {
        const float yr = fminf(fmaxf(f.yR, a0), a1), yl = fminf(fmaxf(-f.yR, a0), a1);
        xmax = f.ecx + f.k * yr + foot_sqrt(fmaxf(f.det * (f.syy - yr * yr), 0.f)) * f.isyy + 1e-3f;
        xmin = f.ecx + f.k * yl - foot_sqrt(fmaxf(f.det * (f.syy - yl * yl), 0.f)) * f.isyy - 1e-3f;
    }
    const float dy = fmaxf(fmaxf(Y0 - f.dcy, f.dcy - Y1), 0.f);
    const float h2 = f.r2sq - dy * dy;
    if (h2 >= 0.f) {
        const float h = foot_sqrt(h2) * 1.000001f;
        xmin = fminf(xmin, f.dcx - h); xmax = fmaxf(xmax, f.dcx + h);
    }
    if (f.unbounded) { xmin = -3.0e38f; xmax = 3.0e38f; }
}
// Sub-tiles: a 16x16 tile is a 4x4 grid of 4x4-pixel sub-tiles (id = 4*(y block) + (x block)); one DPP row of a
// forward wave owns one sub-tile.  16-bit mask of the sub-tiles the footprint can touch.
__device__ __forceinline__ unsigned subtile_overlap(const Foot& f, int tile_x0, int tile_y0) {
    unsigned ov = 0;
#pragma unroll
    for (int by = 0; by < 4; by++) {
        float xmin, xmax;
        foot_strip(f, (float)(tile_y0 + 4 * by), (float)(tile_y0 + 4 * by + 3), xmin, xmax);
        xmin -= (float)tile_x0; xmax -= (float)tile_x0;
#pragma unroll
        for (int bx = 0; bx < 4; bx++) ov |= (xmin <= (float)(4 * bx + 3) && xmax >= (float)(4 * bx)) ? (1u << (4 * by + bx)) : 0u;
    }
    return ov;
}
// The same 16 bits in ROW order: bit 4*w + r <-> the sub-tile owned by DPP row r of wave w (thread_pixel), i.e. bit (tid >> 4).
__device__ __forceinline__ unsigned subtile_overlap_rows(const Foot& f, int tile_x0, int tile_y0) {
    unsigned ov = 0;
#pragma unroll
    for (int by = 0; by < 4; by++) {
        float xmin, xmax;
        foot_strip(f, (float)(tile_y0 + 4 * by), (float)(tile_y0 + 4 * by + 3), xmin, xmax);
        xmin -= (float)tile_x0; xmax -= (float)tile_x0;
#pragma unroll
        for (int bx = 0; bx < 4; bx++) {
            const int rowbit = 4 * (((by >> 1) << 1) | (bx >> 1)) + (((by & 1) << 1) | (bx & 1));
            ov |= (xmin <= (float)(4 * bx + 3) && xmax >= (float)(4 * bx)) ? (1u << rowbit) : 0u;
        }
    }
    return ov;
}
// 4-bit mask of the tile's 8x8 quads (= backward waves; quad id = 2*(y half) + (x half)) the footprint can touch.
__device__ __forceinline__ unsigned quad_overlap(const Foot& f, int tile_x0, int tile_y0) {
    unsigned ov = 0;
#pragma unroll
    for (int by = 0; by < 2; by++) {
        float xmin, xmax;
        foot_strip(f, (float)(tile_y0 + 8 * by), (float)(tile_y0 + 8 * by + 7), xmin, xmax);
        xmin -= (float)tile_x0; xmax -= (float)tile_x0;
#pragma unroll
        for (int bx = 0; bx < 2; bx++) ov |= (xmin <= (float)(8 * bx + 7) && xmax >= (float)(8 * bx)) ? (1u << (2 * by + bx)) : 0u;
    }
    return ov;
}

__device__ __forceinline__ unsigned long long uniform_u64(unsigned long long v) {
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v);
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return ((unsigned long long)hi << 32) | lo;
}

// pixel owned by a thread: wave w -> 8x8 quad (w&1, w>>1); DPP row r = lane>>4 -> 4x4 sub-tile (r&1, r>>1) of the
// quad; lane i = lane&15 -> pixel (i&3, i>>2) of the sub-tile.  `sub` = the sub-tile's id (subtile_overlap bit).
__device__ __forceinline__ void thread_pixel(int tid, int& lx, int& ly, int& sub) {
    const int w = tid >> 6, r = (tid >> 4) & 3, i = tid & 15;
    const int bx = ((w & 1) << 1) | (r & 1), by = (w & 2) | (r >> 1);
    lx = (bx << 2) + (i & 3);
    ly = (by << 2) + (i >> 2);
    sub = (by << 2) | bx;
}

}  // namespace surfel

namespace surfel {
// P = world2ndc * ndc2pix, row r / column c at Pm[3*r+c]; projmatrix = full_proj_transform as torch
// stores it; pixel-centre convention ((ndc+1)*W-1)/2 (gaussian_renderer/__init__.py:69-74).
__device__ __forceinline__ void world2pix(const float* __restrict__ pm, int W, int H, float* Pm) {
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const float m0 = pm[4 * r + 0], m1 = pm[4 * r + 1], m3 = pm[4 * r + 3];
        Pm[3 * r + 0] = m0 * (0.5f * (float)W) + m3 * (0.5f * (float)(W - 1));
        Pm[3 * r + 1] = m1 * (0.5f * (float)H) + m3 * (0.5f * (float)(H - 1));
        Pm[3 * r + 2] = m3;
    }
}
// ---- LDS-DMA (global -> LDS without a VGPR, asynchronous, completion by vmcnt) ----
__device__ __forceinline__ unsigned lds_offset(const void* p) {
    return (unsigned)(uintptr_t)(const __attribute__((address_space(3))) void*)p;
}
// 16 B per lane, global -> LDS at (wave-uniform) lds_base + 16 * lane.  M0 carries the LDS base and is compiler-reserved: saved and
// restored inside the statement.  hipcc does not count this load: the issuer waits with an explicit s_waitcnt vmcnt.
__device__ __forceinline__ void dma16(const void* gsrc, unsigned lds_base) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_base) : "memory");
}
__device__ __forceinline__ void dma4(const void* gsrc, unsigned lds_base) {      // 4 B per lane -> lds_base + 4 * lane
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_base) : "memory");
}

}  // namespace surfel
