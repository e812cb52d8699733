// train_loss_body.h — the L1 + SSIM kernels' bodies as device functions over a caller-provided LDS workspace, so that they can run
// as ordinary kernels (train_loss.hip) and as one half of the horizontally fused training-loss launches (train_fused.hip).
#pragma once
#include <hip/hip_runtime.h>

#include "surfel_common.h"

namespace surfel {
namespace lossk {

constexpr int ST = 32;             // output tile edge
// per window radius SR: SHALO = staged tile edge (42 for SR = 5), NSTAGE = staged elements per thread (7), NE = consecutive
// elements a thread slides its (2 SR + 1)-tap window over to produce 4 outputs (14), XP = float2 pitch of the staged tile
// (even: 16-B aligned b128 reads at even columns)
#define SSIM_GEOMETRY(SR)                                                                                           \
    constexpr int SHALO = ST + 2 * (SR), NSTAGE = (SHALO * SHALO + 255) / 256, NE = 4 + 2 * (SR), NW = 2 * (SR) + 1, \
                  XP = SHALO + 4
struct SsimWin { float w[15]; };     // the normalised 1-D Gaussian window, passed by value
constexpr float SSIM_C1 = 0.01f * 0.01f;
constexpr float SSIM_C2 = 0.03f * 0.03f;

// gaussian(11, 1.5) of loss_utils.py:29-31 evaluated in fp32 exactly as torch does (exp in double, stored fp32, fp32 sum)
constexpr float kG11[11] = {1.028380124e-03f, 7.598758209e-03f, 3.600077331e-02f, 1.093606874e-01f,
                            2.130055279e-01f, 2.660117149e-01f, 2.130055279e-01f, 1.093606874e-01f,
                            3.600077331e-02f, 7.598758209e-03f, 1.028380124e-03f};

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// LDS layout (27.6 KB at window 11 -> 5 workgroups / CU): the staged tile as float2 (x,y) with pitch XP; the horizontal moments as
// float4 (E[x], E[y], E[x^2], E[y^2]) with pitch HZP + float (E[xy]) are written OVER it — every thread keeps its horizontal-pass
// results in registers across the barrier that retires the staged tile.  Each thread slides the window over NE consecutive elements
// held in registers (window 11: 4 outputs per 14 LDS reads instead of 44), horizontally then vertically.
// (Round 5, scripts/loss_trace.hip: at 43 KB = 3 workgroups / CU a workgroup took 8.2 us for ~3.7 us of vector issue and the launch
// was the sum of its workgroups' latencies over 768 slots.)
constexpr int HZP = ST + 1;       // float4 pitch of the horizontal-pass result

// Horizontal moments of NO consecutive outputs of one staged row: the (2 SR + 1)-tap window slid over NO + 2 SR elements.  The taps of
// an output are accumulated in ascending order whatever NO is: same bits for every split of a row into items.
template <int SR, int NO>
__device__ __forceinline__ void hz_moments5(const float2* __restrict__ row, const SsimWin& win, float (&a)[NO], float (&b)[NO], float (&aa)[NO], float (&bb)[NO],
                                            float (&ab)[NO]) {
    constexpr int NEL = NO + 2 * SR, NW = 2 * SR + 1;
    float xv[NEL], yv[NEL];
    const float4* src = reinterpret_cast<const float4*>(row);
#pragma unroll
    for (int k = 0; k < NEL / 2; k++) { const float4 t = src[k]; xv[2 * k] = t.x; yv[2 * k] = t.y; xv[2 * k + 1] = t.z; yv[2 * k + 1] = t.w; }
#pragma unroll
    for (int j = 0; j < NO; j++) { a[j] = 0.f; b[j] = 0.f; aa[j] = 0.f; bb[j] = 0.f; ab[j] = 0.f; }
#pragma unroll
    for (int e = 0; e < NEL; e++) {
        const float x = xv[e], y = yv[e], xx = x * x, yy = y * y, xy = x * y;
#pragma unroll
        for (int j = 0; j < NO; j++) {
            const int k = e - j;
            if (k >= 0 && k < NW) {
                const float w = win.w[k];
                a[j] += w * x; b[j] += w * y; aa[j] += w * xx; bb[j] += w * yy; ab[j] += w * xy;
            }
        }
    }
}

template <int SR>
__device__ __forceinline__ void ssim_fwd_body(char* smem, int vblock, int vgrid, int H, int W, const float* __restrict__ img, const float* __restrict__ gt,
                                                       float* __restrict__ dmaps, size_t map_stride, float* __restrict__ partials, SsimWin win) {
    SSIM_GEOMETRY(SR);
    // LDS carve-up (the caller provides ssim_fwd_lds<SR>() bytes, 16-B aligned): sxy, and over it after the horizontal pass hz4 | hz1 | red
    float2* const sxy = reinterpret_cast<float2*>(smem);
    float4* const hz4 = reinterpret_cast<float4*>(smem);
    float* const hz1 = reinterpret_cast<float*>(smem + sizeof(float4) * SHALO * HZP);
    float* const red = hz1 + SHALO * ST;
    const int tid = threadIdx.x;
    // workgroup b runs on XCD b % 8: every XCD gets a contiguous run of (plane, tile) so that neighbouring tiles' halos hit in
    // one L2 instead of being fetched from HBM once per XCD
    const int gxt = (W + ST - 1) / ST, ntile = gxt * ((H + ST - 1) / ST);
    const int lin = xcd_tile(vblock, vgrid);
    const int plane = lin / ntile, tile = lin - plane * ntile;
    const int x0 = (tile % gxt) * ST, y0 = (tile / gxt) * ST;
    const size_t poff = (size_t)plane * H * W;
    const float* X = img + poff;
    const float* Y = gt + poff;
    {   // stage the tile (42x42 at window 11): ALL of a thread's 7 x 2 loads are issued before the first one is consumed (as a rolled loop this
        // was seven dependent global round trips per workgroup — most of the kernel's time)
        float2 v[NSTAGE];
#pragma unroll
        for (int k = 0; k < NSTAGE; k++) {
            const int i = tid + 256 * k;
            const int r = i / SHALO, c = i - r * SHALO;
            const int gy = y0 + r - SR, gx = x0 + c - SR;
            const bool in = i < SHALO * SHALO && gy >= 0 && gy < H && gx >= 0 && gx < W;
            v[k] = in ? make_float2(X[(size_t)gy * W + gx], Y[(size_t)gy * W + gx]) : make_float2(0.f, 0.f);
        }
#pragma unroll
        for (int k = 0; k < NSTAGE; k++) {
            const int i = tid + 256 * k;
            const int r = i / SHALO, c = i - r * SHALO;
            if (i < SHALO * SHALO) sxy[r * XP + c] = v[k];
        }
    }
    __syncthreads();
    // horizontal pass into registers.  Staged rows 0 .. 31: one item of 4 output columns per thread (256 items); the 2 SR rows behind
    // them: items of 2 output columns (16 per row: 160 at window 11) — as 80 four-column items they were a second trip of the first
    // 80 threads only, i.e. twice the work for the SIMDs that hold waves 0 and 1
    const int r1 = tid >> 3, c1 = (tid & 7) << 2;
    const int r2 = ST + (tid >> 4), c2 = (tid & 15) << 1;
    const bool has2 = tid < 2 * SR * 16;
    float a1[4], b1[4], aa1[4], bb1[4], ab1[4], a2[2], b2[2], aa2[2], bb2[2], ab2[2];
    hz_moments5<SR, 4>(&sxy[r1 * XP + c1], win, a1, b1, aa1, bb1, ab1);
    // The first item's results are pinned here: left alone, the compiler sinks its ~250 multiply-adds below the second item (towards
    // the stores behind the barrier) and keeps BOTH items' inputs live — 25 registers and the fifth workgroup per CU.
#pragma unroll
    for (int j = 0; j < 4; j++) asm volatile("" : "+v"(a1[j]), "+v"(b1[j]), "+v"(aa1[j]), "+v"(bb1[j]), "+v"(ab1[j]));
    if (has2) hz_moments5<SR, 2>(&sxy[r2 * XP + c2], win, a2, b2, aa2, bb2, ab2);
    const int c = tid & 31, g = tid >> 5;      // vertical pass: thread -> column c, output rows 4g .. 4g+3
    float l1 = 0.f;                            // ... whose own pixels' L1 term is taken now: the staged tile is about to be overwritten
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const float2 v = sxy[(4 * g + j + SR) * XP + c + SR];
        if (y0 + 4 * g + j < H && x0 + c < W) l1 += fabsf(v.x - v.y);
    }
    __syncthreads();                           // the staged tile is dead: the moments go over it
#pragma unroll
    for (int j = 0; j < 4; j++) {
        hz4[r1 * HZP + c1 + j] = make_float4(a1[j], b1[j], aa1[j], bb1[j]);
        hz1[r1 * ST + c1 + j] = ab1[j];
    }
    if (has2) {
#pragma unroll
        for (int j = 0; j < 2; j++) {
            hz4[r2 * HZP + c2 + j] = make_float4(a2[j], b2[j], aa2[j], bb2[j]);
            hz1[r2 * ST + c2 + j] = ab2[j];
        }
    }
    __syncthreads();
    float mu1[4] = {0.f, 0.f, 0.f, 0.f}, mu2[4] = {0.f, 0.f, 0.f, 0.f}, e11[4] = {0.f, 0.f, 0.f, 0.f}, e22[4] = {0.f, 0.f, 0.f, 0.f},
          e12[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int e = 0; e < NE; e++) {
        if (e == NE / 2) {      // the second half's 35 LDS values are requested behind the first half's arithmetic (all 70 in flight: spills at 96 registers)
#pragma unroll
            for (int j = 0; j < 4; j++) asm volatile("" : "+v"(mu1[j]), "+v"(mu2[j]), "+v"(e11[j]), "+v"(e22[j]), "+v"(e12[j]) :: "memory");
        }
        const float4 h = hz4[(4 * g + e) * HZP + c];
        const float h1 = hz1[(4 * g + e) * ST + c];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int k = e - j;
            if (k >= 0 && k < NW) {
                const float w = win.w[k];
                mu1[j] += w * h.x; mu2[j] += w * h.y; e11[j] += w * h.z; e22[j] += w * h.w; e12[j] += w * h1;
            }
        }
    }
    float ss = 0.f;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        // One rounding per operation as written, like the reference's tensor ops (loss_utils.py:46-61): under -ffp-contract=fast the
        // compiler contracts "ss += A * B * iCD" or "x * y + z * w" one way or the other depending on the code AROUND them (seen in
        // round 5: moving the L1 term out of this loop changed the bits of ss), and the fused and the plain launches must agree.
#pragma clang fp contract(off)
        const int r = 4 * g + j;
        const int gy = y0 + r, gx = x0 + c;
        if (gy < H && gx < W) {
            const float mu1_sq = mu1[j] * mu1[j], mu2_sq = mu2[j] * mu2[j], mu12 = mu1[j] * mu2[j];
            const float s1 = e11[j] - mu1_sq, s2 = e22[j] - mu2_sq, s12 = e12[j] - mu12;
            const float A = 2.f * mu12 + SSIM_C1, B = 2.f * s12 + SSIM_C2;
            const float C = mu1_sq + mu2_sq + SSIM_C1, D = s1 + s2 + SSIM_C2;
            const float iCD = 1.f / (C * D);
            const float S = A * B * iCD;
            ss += S;
            if (dmaps) {
                const size_t o = poff + (size_t)gy * W + gx;
                dmaps[o] = 2.f * mu2[j] * (B - A) * iCD + 2.f * mu1[j] * S * (1.f / D - 1.f / C);   // dS/dmu1 (mu1, E[x^2], E[xy] independent)
                dmaps[map_stride + o] = -S / D;                                                   // dS/dE[x^2]
                dmaps[2 * map_stride + o] = 2.f * A * iCD;                                        // dS/dE[xy]
            }
        }
    }
    l1 = wave_sum(l1); ss = wave_sum(ss);
    if ((tid & 63) == 0) { red[2 * (tid >> 6)] = l1; red[2 * (tid >> 6) + 1] = ss; }
    __syncthreads();
    if (tid == 0) {
        const size_t blk = (size_t)lin;
        partials[2 * blk] = (red[0] + red[2]) + (red[4] + red[6]);
        partials[2 * blk + 1] = (red[1] + red[3]) + (red[5] + red[7]);
    }
}


// the backward's horizontal pass: three maps, NO consecutive outputs of one staged row (taps in ascending order: hz_moments5)
template <int SR, int NO>
__device__ __forceinline__ void hz_moments3(const float2* __restrict__ row12, const float* __restrict__ row3, const SsimWin& win, float (&a)[NO], float (&b)[NO],
                                            float (&d)[NO]) {
    constexpr int NEL = NO + 2 * SR, NW = 2 * SR + 1;
    float m1[NEL], m2[NEL], m3[NEL];
    const float4* src = reinterpret_cast<const float4*>(row12);
#pragma unroll
    for (int k = 0; k < NEL / 2; k++) { const float4 t = src[k]; m1[2 * k] = t.x; m2[2 * k] = t.y; m1[2 * k + 1] = t.z; m2[2 * k + 1] = t.w; }
    const float2* src3 = reinterpret_cast<const float2*>(row3);
#pragma unroll
    for (int k = 0; k < NEL / 2; k++) { const float2 t = src3[k]; m3[2 * k] = t.x; m3[2 * k + 1] = t.y; }
#pragma unroll
    for (int j = 0; j < NO; j++) { a[j] = 0.f; b[j] = 0.f; d[j] = 0.f; }
#pragma unroll
    for (int e = 0; e < NEL; e++) {
#pragma unroll
        for (int j = 0; j < NO; j++) {
            const int k = e - j;
            if (k >= 0 && k < NW) { const float w = win.w[k]; a[j] += w * m1[e]; b[j] += w * m2[e]; d[j] += w * m3[e]; }
        }
    }
}

template <int SR>
__device__ __forceinline__ void ssim_bwd_body(char* smem, int vblock, int vgrid, int H, int W, const float* __restrict__ img, const float* __restrict__ gt,
                                                       const float* __restrict__ dmaps, size_t map_stride, float c_l1, float c_ssim,
                                                       const float* __restrict__ g_l1_dev, const float* __restrict__ g_ssim_dev,
                                                       float* __restrict__ grad_img, SsimWin win) {
    SSIM_GEOMETRY(SR);
    // LDS carve-up (ssim_bwd_lds<SR>() bytes): s12 (M1, M2) | s3 (M3), and over them after the horizontal pass hz = (M1, M2, M3, -) per
    // element (the results wait in registers across the barrier, as in the forward body: 23 KB instead of 45)
    float4* const hz = reinterpret_cast<float4*>(smem);
    float2* const s12 = reinterpret_cast<float2*>(smem);
    float* const s3 = reinterpret_cast<float*>(smem + sizeof(float2) * SHALO * XP);
    const int tid = threadIdx.x;
    // workgroup b runs on XCD b % 8: every XCD gets a contiguous run of (plane, tile) so that neighbouring tiles' halos hit in
    // one L2 instead of being fetched from HBM once per XCD
    const int gxt = (W + ST - 1) / ST, ntile = gxt * ((H + ST - 1) / ST);
    const int lin = xcd_tile(vblock, vgrid);
    const int plane = lin / ntile, tile = lin - plane * ntile;
    const int x0 = (tile % gxt) * ST, y0 = (tile / gxt) * ST;
    const size_t poff = (size_t)plane * H * W;
    {   // all 7 x 3 loads of a thread in flight before the first LDS store (see ssim_fwd_kernel)
        float2 v12[NSTAGE];
        float v3[NSTAGE];
#pragma unroll
        for (int k = 0; k < NSTAGE; k++) {
            const int i = tid + 256 * k;
            const int r = i / SHALO, c = i - r * SHALO;
            const int gy = y0 + r - SR, gx = x0 + c - SR;
            const bool in = i < SHALO * SHALO && gy >= 0 && gy < H && gx >= 0 && gx < W;
            const size_t o = poff + (size_t)gy * W + gx;
            v12[k] = in ? make_float2(dmaps[o], dmaps[map_stride + o]) : make_float2(0.f, 0.f);
            v3[k] = in ? dmaps[2 * map_stride + o] : 0.f;
        }
#pragma unroll
        for (int k = 0; k < NSTAGE; k++) {
            const int i = tid + 256 * k;
            const int r = i / SHALO, c = i - r * SHALO;
            if (i < SHALO * SHALO) { s12[r * XP + c] = v12[k]; s3[r * XP + c] = v3[k]; }
        }
    }
    __syncthreads();
    // horizontal pass into registers: 4 output columns per thread for staged rows 0 .. 31, 2 per thread for the 2 SR rows behind (ssim_fwd_body)
    const int r1 = tid >> 3, c1 = (tid & 7) << 2;
    const int r2 = ST + (tid >> 4), c2 = (tid & 15) << 1;
    const bool has2 = tid < 2 * SR * 16;
    float a1[4], b1[4], d1[4], a2[2], b2[2], d2[2];
    hz_moments3<SR, 4>(&s12[r1 * XP + c1], &s3[r1 * XP + c1], win, a1, b1, d1);
#pragma unroll
    for (int j = 0; j < 4; j++) asm volatile("" : "+v"(a1[j]), "+v"(b1[j]), "+v"(d1[j]));      // (pinned: see ssim_fwd_body)
    if (has2) hz_moments3<SR, 2>(&s12[r2 * XP + c2], &s3[r2 * XP + c2], win, a2, b2, d2);
    __syncthreads();                           // the staged maps are dead: the horizontal pass goes over them
#pragma unroll
    for (int j = 0; j < 4; j++) hz[r1 * HZP + c1 + j] = make_float4(a1[j], b1[j], d1[j], 0.f);
    if (has2) {
#pragma unroll
        for (int j = 0; j < 2; j++) hz[r2 * HZP + c2 + j] = make_float4(a2[j], b2[j], d2[j], 0.f);
    }
    __syncthreads();
    const float k_l1 = c_l1 * (g_l1_dev ? g_l1_dev[0] : 1.f), k_ss = c_ssim * (g_ssim_dev ? g_ssim_dev[0] : 1.f);
    const int c = tid & 31, g = tid >> 5;
    float a[4] = {0.f, 0.f, 0.f, 0.f}, b[4] = {0.f, 0.f, 0.f, 0.f}, d[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int e = 0; e < NE; e++) {
        if (e == NE / 2) {      // (second half of the LDS reads behind the first half's arithmetic: ssim_fwd_body)
#pragma unroll
            for (int j = 0; j < 4; j++) asm volatile("" : "+v"(a[j]), "+v"(b[j]), "+v"(d[j]) :: "memory");
        }
        const float4 h = hz[(4 * g + e) * HZP + c];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int k = e - j;
            if (k >= 0 && k < NW) { const float w = win.w[k]; a[j] += w * h.x; b[j] += w * h.y; d[j] += w * h.z; }
        }
    }
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int gy = y0 + 4 * g + j, gx = x0 + c;
        if (gy >= H || gx >= W) continue;
        const size_t o = poff + (size_t)gy * W + gx;
        const float xv = img[o], yv = gt[o];
        const float df = xv - yv;
        const float sgn = df > 0.f ? 1.f : (df < 0.f ? -1.f : 0.f);
        // (spelled out: "a * b + c * d" leaves the compiler two ways to contract, and it takes either depending on the code around it)
        const float conv = __builtin_fmaf(yv, d[j], __builtin_fmaf(2.f * xv, b[j], a[j]));
        grad_img[o] = __builtin_fmaf(k_ss, conv, k_l1 * sgn);
    }
}


// All loss scalars of an iteration from the two partial-sum arrays (train.py:72-88), fixed summation order — that of 1024 threads
// (strided partial sums, wave totals by xor-shuffles, the 16 wave totals added in wave order) whatever the calling workgroup's size
// NT (1024: loss_finalize_kernel; 256: the extra workgroup of the fused backward plays four of those waves each): same bits.
// out = [Ll1, ssim, normal_err, dist, photometric, total].  pb may be NULL (no regularisers).  red: 4 x 16 floats of LDS.
struct LossFinalize {
    const float* pa; int na; float scale_a; const float* pb; int nb; float scale_b;
    float lambda_dssim, lambda_normal, lambda_dist; float* out; float* total_out;
};

template <int NT>
__device__ __forceinline__ void loss_finalize_body(float (*red)[16], const LossFinalize& f) {
    const int tid = threadIdx.x;
    const float2* __restrict__ pa2 = reinterpret_cast<const float2*>(f.pa);
    const float2* __restrict__ pb2 = reinterpret_cast<const float2*>(f.pb);
#pragma unroll
    for (int j = 0; j < 1024 / NT; j++) {
        const int vt = tid + NT * j;
        float a0 = 0.f, a1 = 0.f, b0 = 0.f, b1 = 0.f;
        for (int i = vt; i < f.na; i += 1024) { const float2 v = pa2[i]; a0 += v.x; a1 += v.y; }
        if (f.pb) for (int i = vt; i < f.nb; i += 1024) { const float2 v = pb2[i]; b0 += v.x; b1 += v.y; }
        a0 = wave_sum(a0); a1 = wave_sum(a1); b0 = wave_sum(b0); b1 = wave_sum(b1);
        if ((tid & 63) == 0) { red[0][vt >> 6] = a0; red[1][vt >> 6] = a1; red[2][vt >> 6] = b0; red[3][vt >> 6] = b1; }
    }
    __syncthreads();
    if (tid == 0) {
        float t[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            float acc = 0.f;
#pragma unroll
            for (int w = 0; w < 16; w++) acc += red[k][w];
            t[k] = acc;
        }
        const float l1 = t[0] * f.scale_a, ss = t[1] * f.scale_a, ne = t[2] * f.scale_b, di = t[3] * f.scale_b;
        const float ph = (1.f - f.lambda_dssim) * l1 + f.lambda_dssim * (1.f - ss);
        f.out[0] = l1; f.out[1] = ss; f.out[2] = ne; f.out[3] = di; f.out[4] = ph;
        const float tot = ph + f.lambda_normal * ne + f.lambda_dist * di;
        f.out[5] = tot;
        if (f.total_out) f.total_out[0] = tot;
    }
}

constexpr size_t lds_max(size_t a, size_t b) { return a > b ? a : b; }
template <int SR> constexpr size_t ssim_fwd_lds() {
    return lds_max(sizeof(float2) * (ST + 2 * SR) * (ST + 2 * SR + 4), sizeof(float4) * (ST + 2 * SR) * HZP + sizeof(float) * (ST + 2 * SR) * ST + 32);
}
template <int SR> constexpr size_t ssim_bwd_lds() {
    return lds_max((sizeof(float2) + sizeof(float)) * (ST + 2 * SR) * (ST + 2 * SR + 4), sizeof(float4) * (ST + 2 * SR) * HZP);
}

}  // namespace lossk
}  // namespace surfel
