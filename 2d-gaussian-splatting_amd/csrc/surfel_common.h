// surfel_common.h — shared definitions for the gfx950 surfel rasterizer kernels.
// Written for CDNA4 only: wave64, 16x16-pixel tiles = 4 waves, each wave owning an 8x8 pixel quad.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace surfel {

constexpr int TILE = 16;            // tile edge in pixels (fixed by the reference's binning semantics)
constexpr int BLOCK = TILE * TILE;  // 256 threads = 4 waves
constexpr int WAVE = 64;

// rasterizer constants (oracle/surfel_oracle.c holds the same list with provenance)
constexpr float NEAR_N = 0.2f;
constexpr float FAR_N = 100.0f;
constexpr float FILTER_SIZE = 0.707106f;
constexpr float FILTER_INV_SQUARE = 2.0f;
constexpr float CUTOFF = 3.0f;
constexpr float ALPHA_MIN = 1.0f / 255.0f;
constexpr float ALPHA_MAX = 0.99f;
constexpr float T_EPS = 0.0001f;

// Per-surfel packed record, 96 B = 6 x float4, written by preprocess, gathered by the blend kernels.
// One record touches at most two 128-B lines (vs. seven separate arrays in an SoA layout).
//   q0 = Tu.x Tu.y Tu.z Tv.x
//   q1 = Tv.y Tv.z Tw.x Tw.y
//   q2 = Tw.z xy.x xy.y opacity
//   q3 = n.x  n.y  n.z  r
//   q4 = g    b    inst_base(u32 bits)  rect(u32 bits: x0 | y0<<10 | w<<20)
//   q5 = xmin xmax ymin ymax : conservative pixel bbox of {alpha >= 1/255} (cull only, never changes results)
constexpr int REC_F = 24;
constexpr int REC_Q = REC_F / 4;
// Per-(tile,surfel) gradient record written by blend-backward, summed by preprocess-backward:
//   [0..8] dL/dT (Tu,Tv,Tw)  [9..10] dL/dxy (low-pass branch)  [11..13] dL/dnormal  [14] dL/dopacity
//   [15..17] dL/drgb  [18..19] pad
constexpr int GREC_F = 20;

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

// Which of the tile's four 8x8 quads (= waves) can a surfel with pixel bbox (xmin,xmax,ymin,ymax) touch?
__device__ __forceinline__ unsigned quad_overlap(const float4 bb, int tile_x0, int tile_y0) {
    const float x0 = (float)tile_x0, y0 = (float)tile_y0;
    const bool xl = bb.x <= x0 + 7.f && bb.y >= x0, xr = bb.x <= x0 + 15.f && bb.y >= x0 + 8.f;
    const bool yt = bb.z <= y0 + 7.f && bb.w >= y0, yb = bb.z <= y0 + 15.f && bb.w >= y0 + 8.f;
    return (xl && yt ? 1u : 0u) | (xr && yt ? 2u : 0u) | (xl && yb ? 4u : 0u) | (xr && yb ? 8u : 0u);
}

__device__ __forceinline__ unsigned long long uniform_u64(unsigned long long v) {
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v);
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return ((unsigned long long)hi << 32) | lo;
}

// pixel owned by a thread: wave w -> 8x8 quad (w&1, w>>1); lane -> (lane&7, lane>>3)
__device__ __forceinline__ void thread_pixel(int tid, int& lx, int& ly) {
    const int w = tid >> 6, l = tid & 63;
    lx = ((w & 1) << 3) + (l & 7);
    ly = ((w >> 1) << 3) + (l >> 3);
}

struct Hit {
    float sx, sy, pz, kx, ky, kz, lx, ly, lz, dx, dy, depth, G, alpha;
    bool use3d;
};

// Ray–splat intersection and alpha for one (pixel, surfel) pair; false = pair skipped.
__device__ __forceinline__ bool intersect(const float4 q0, const float4 q1, const float4 q2, float pxf, float pyf, Hit& h) {
    const float Tux = q0.x, Tuy = q0.y, Tuz = q0.z, Tvx = q0.w, Tvy = q1.x, Tvz = q1.y;
    const float Twx = q1.z, Twy = q1.w, Twz = q2.x;
    h.kx = pxf * Twx - Tux; h.ky = pxf * Twy - Tuy; h.kz = pxf * Twz - Tuz;
    h.lx = pyf * Twx - Tvx; h.ly = pyf * Twy - Tvy; h.lz = pyf * Twz - Tvz;
    const float p0 = h.ky * h.lz - h.kz * h.ly;
    const float p1 = h.kz * h.lx - h.kx * h.lz;
    const float p2 = h.kx * h.ly - h.ky * h.lx;
    if (p2 == 0.0f) return false;
    h.pz = p2;
    const float ip = __builtin_amdgcn_rcpf(p2);
    h.sx = p0 * ip; h.sy = p1 * ip;
    const float rho3d = h.sx * h.sx + h.sy * h.sy;
    h.dx = q2.y - pxf; h.dy = q2.z - pyf;
    const float rho2d = FILTER_INV_SQUARE * (h.dx * h.dx + h.dy * h.dy);
    h.use3d = rho3d <= rho2d;
    const float rho = fminf(rho3d, rho2d);
    h.depth = h.use3d ? (h.sx * Twx + h.sy * Twy) + Twz : Twz;
    if (h.depth < NEAR_N) return false;
    const float power = -0.5f * rho;
    if (power > 0.0f) return false;
    h.G = __expf(power);
    h.alpha = fminf(ALPHA_MAX, q2.w * h.G);
    return h.alpha >= ALPHA_MIN;
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
}  // namespace surfel
