// train_post_body.h — the allmap post-processing kernels' bodies as device functions over a caller-provided LDS workspace
// (train_post.hip: ordinary kernels; train_fused.hip: one half of the horizontally fused training-loss launches).
#pragma once
#include <hip/hip_runtime.h>

#include <cfloat>

#include "surfel_common.h"

namespace surfel {
namespace postk {

constexpr int PT = 16;

struct Cam {
    float A[9], K[9], o[3];
};
__device__ __forceinline__ Cam load_cam(const float* __restrict__ cam) {
    Cam c;
#pragma unroll
    for (int i = 0; i < 9; i++) { c.A[i] = cam[i]; c.K[i] = cam[9 + i]; }
    c.o[0] = cam[18]; c.o[1] = cam[19]; c.o[2] = cam[20];
    return c;
}

// torch.nan_to_num(x, 0, 0): nan -> 0, +inf -> 0, -inf -> lowest finite (gaussian_renderer/__init__.py:127,132)
__device__ __forceinline__ float nan_to_num00(float x) {
    if (x != x) return 0.f;
    if (x == INFINITY) return 0.f;
    if (x == -INFINITY) return -FLT_MAX;
    return x;
}
__device__ __forceinline__ bool finite_f(float x) { return fabsf(x) <= FLT_MAX; }   // false for nan / inf

// One rounding per operation as written in this file's per-pixel arithmetic (#pragma clang fp contract(off) in every function): the
// expressions are full of "a * b + c * d" (cross products, ray directions, blends), which -ffp-contract=fast contracts one way or the
// other depending on the code around them — and the fused and the plain launches, two inlinings of these bodies, must give the same
// bits.  It is also what the reference's tensor ops do (utils/point_utils.py:6-36: separate multiplies, adds, torch.cross).
__device__ __forceinline__ float surf_depth_of(float a0, float a1, float a5, float ratio) {
#pragma clang fp contract(off)
    const float expd = nan_to_num00(a0 / a1);
    const float med = nan_to_num00(a5);
    return expd * (1.f - ratio) + ratio * med;
}
__device__ __forceinline__ float surf_depth_at(const float* __restrict__ allmap, size_t HW, size_t o, float ratio) {
    return surf_depth_of(allmap[o], allmap[HW + o], allmap[5 * HW + o], ratio);
}

__device__ __forceinline__ void ray_dir(const Cam& c, int x, int y, float* d) {
#pragma clang fp contract(off)
    const float fx = (float)x, fy = (float)y;
#pragma unroll
    for (int j = 0; j < 3; j++) d[j] = fx * c.K[j] + fy * c.K[3 + j] + c.K[6 + j];
}

__device__ __forceinline__ void cross3(const float* a, const float* b, float* r) {
#pragma clang fp contract(off)
    r[0] = a[1] * b[2] - a[2] * b[1];
    r[1] = a[2] * b[0] - a[0] * b[2];
    r[2] = a[0] * b[1] - a[1] * b[0];
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// maps: 0 alpha | 1-3 rend_normal | 4 dist | 5 surf_depth | 6-8 surf_normal
__device__ __forceinline__ void post_fwd_body(char* smem, int vblock, int vgrid, int H, int W, const float* __restrict__ allmap, const float* __restrict__ cam,
                                                       float ratio, float* __restrict__ maps, float* __restrict__ partials) {
#pragma clang fp contract(off)
    constexpr int HB = PT + 2;   // 18
    float* const px = reinterpret_cast<float*>(smem);      // LDS carve-up (post_fwd_lds() bytes): px | py | pz | red
    float* const py = px + HB * HB;
    float* const pz = py + HB * HB;
    float* const red = pz + HB * HB;
    const Cam c = load_cam(cam);
    const int tid = threadIdx.x;
    // workgroup b runs on XCD b % 8: give every XCD a contiguous band of tiles so that the halos neighbouring tiles share are
    // served by one L2 instead of being fetched from HBM once per XCD (measured 2.4x the algorithmic bytes without this)
    const int gxt = (W + PT - 1) / PT;
    const int tile = xcd_tile(vblock, vgrid);
    const int x0 = (tile % gxt) * PT, y0 = (tile / gxt) * PT;
    const size_t HW = (size_t)H * W;
    // The thread's own pixel first: its six values are requested in front of the staging loop, so that the workgroup pays ONE global
    // round trip instead of two (a workgroup is all latency: scripts/loss_trace.hip, 2.9 us for 256 pixels).
    const int lx = tid & 15, ly = tid >> 4;
    const int gx = x0 + lx, gy = y0 + ly;
    const bool inside = gx < W && gy < H;
    const size_t o = inside ? (size_t)gy * W + gx : 0;
    float a0 = 0.f, alpha = 0.f, a5 = 0.f, dist = 0.f, nv[3] = {0.f, 0.f, 0.f};
    if (inside) {
        a0 = allmap[o]; alpha = allmap[HW + o]; a5 = allmap[5 * HW + o]; dist = allmap[6 * HW + o];
        nv[0] = allmap[2 * HW + o]; nv[1] = allmap[3 * HW + o]; nv[2] = allmap[4 * HW + o];
    }
    for (int i = tid; i < HB * HB; i += 256) {
        const int r = i / HB, cc = i - r * HB;
        const int qy = y0 + r - 1, qx = x0 + cc - 1;
        float p[3] = {0.f, 0.f, 0.f};
        if (qy >= 0 && qy < H && qx >= 0 && qx < W) {
            const float sd = surf_depth_at(allmap, HW, (size_t)qy * W + qx, ratio);
            float d[3];
            ray_dir(c, qx, qy, d);
#pragma unroll
            for (int j = 0; j < 3; j++) p[j] = sd * d[j] + c.o[j];
        }
        px[i] = p[0]; py[i] = p[1]; pz[i] = p[2];
    }
    __syncthreads();
    float e_n = 0.f, e_d = 0.f;
    if (inside) {
        float rn[3];
#pragma unroll
        for (int j = 0; j < 3; j++) rn[j] = c.A[3 * j] * nv[0] + c.A[3 * j + 1] * nv[1] + c.A[3 * j + 2] * nv[2];
        float sn[3] = {0.f, 0.f, 0.f};
        if (gx >= 1 && gx <= W - 2 && gy >= 1 && gy <= H - 2) {
            const int q = (ly + 1) * HB + lx + 1;
            const float dxv[3] = {px[q + HB] - px[q - HB], py[q + HB] - py[q - HB], pz[q + HB] - pz[q - HB]};   // rows y+1 / y-1
            const float dyv[3] = {px[q + 1] - px[q - 1], py[q + 1] - py[q - 1], pz[q + 1] - pz[q - 1]};         // columns x+1 / x-1
            float v[3];
            cross3(dxv, dyv, v);
            const float len = sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
            const float inv = 1.f / fmaxf(len, 1e-12f);
#pragma unroll
            for (int j = 0; j < 3; j++) sn[j] = v[j] * inv * alpha;
        }
        const float sd = surf_depth_of(a0, alpha, a5, ratio);      // recomputed: recovering it from the staged point would lose bits
        if (maps) {      // NULL: only the regulariser sums are wanted (the training loss: 36 B/pixel of stores saved)
            maps[o] = alpha;
            maps[HW + o] = rn[0]; maps[2 * HW + o] = rn[1]; maps[3 * HW + o] = rn[2];
            maps[4 * HW + o] = dist;
            maps[5 * HW + o] = sd;
            maps[6 * HW + o] = sn[0]; maps[7 * HW + o] = sn[1]; maps[8 * HW + o] = sn[2];
        }
        e_n = 1.f - (rn[0] * sn[0] + rn[1] * sn[1] + rn[2] * sn[2]);
        e_d = dist;
    }
    if (partials) {
        e_n = wave_sum(e_n); e_d = wave_sum(e_d);
        if ((tid & 63) == 0) { red[2 * (tid >> 6)] = e_n; red[2 * (tid >> 6) + 1] = e_d; }
        __syncthreads();
        if (tid == 0) {
            const size_t blk = (size_t)tile;
            partials[2 * blk] = (red[0] + red[2]) + (red[4] + red[6]);
            partials[2 * blk + 1] = (red[1] + red[3]) + (red[5] + red[7]);
        }
    }
}


__device__ __forceinline__ void post_bwd_body(char* smem, int vblock, int vgrid, int H, int W, const float* __restrict__ allmap, const float* __restrict__ cam,
                                                       float ratio, const float* __restrict__ gmaps, float c_normal, float c_dist,
                                                       const float* __restrict__ gscale_dev, float* __restrict__ gall) {
#pragma clang fp contract(off)
    constexpr int HP = PT + 4;   // 20: points, 2-pixel halo
    constexpr int HD = PT + 2;   // 18: per-pixel normal gradients, 1-pixel halo
    constexpr int ND = (HD * HD + 255) / 256;      // normal-gradient items per thread (2)
    // LDS carve-up (post_bwd_lds() bytes): px | py | pz | dd: d(loss)/d(dx vector) [0..2], d(loss)/d(dy vector) [3..5] of the pixel's
    // normal | sn_s: the pixel's surf_normal (unit normal * alpha) for the fused regulariser
    float* const px = reinterpret_cast<float*>(smem);
    float* const py = px + HP * HP;
    float* const pz = py + HP * HP;
    float (*const dd)[HD * HD] = reinterpret_cast<float (*)[HD * HD]>(pz + HP * HP);
    float (*const sn_s)[HD * HD] = reinterpret_cast<float (*)[HD * HD]>(pz + HP * HP + 6 * HD * HD);
    const Cam c = load_cam(cam);
    const int tid = threadIdx.x;
    const int gxt = (W + PT - 1) / PT;
    const int tile = xcd_tile(vblock, vgrid);      // XCD-contiguous tile bands (see post_fwd_kernel)
    const int x0 = (tile % gxt) * PT, y0 = (tile / gxt) * PT;
    const size_t HW = (size_t)H * W;
    const float gs = gscale_dev ? gscale_dev[0] : 1.f;
    const float cn = c_normal * gs, cd = c_dist * gs;
    // Everything the later phases read from global memory is requested HERE, in front of the staging loop: the workgroup then pays one
    // global round trip instead of three dependent ones (staging -> barrier -> normals' alpha / rendered normal -> barrier -> the own
    // pixel's depth channels; scripts/loss_trace.hip: 4.7 us per workgroup, all of it latency).
    const int lx = tid & 15, ly = tid >> 4;
    const int gx = x0 + lx, gy = y0 + ly;
    const bool inside = gx < W && gy < H;
    const size_t o = inside ? (size_t)gy * W + gx : 0;
    float a0 = 0.f, a1 = 0.f, a5 = 0.f;
    if (inside) { a0 = allmap[o]; a1 = allmap[HW + o]; a5 = allmap[5 * HW + o]; }
    float alpha_d[ND], nv_d[ND][3];
#pragma unroll
    for (int k = 0; k < ND; k++) {
        const int i = tid + 256 * k;
        const int r = i / HD, cc = i - r * HD;
        const int qy = y0 + r - 1, qx = x0 + cc - 1;
        alpha_d[k] = 0.f; nv_d[k][0] = 0.f; nv_d[k][1] = 0.f; nv_d[k][2] = 0.f;
        if (i < HD * HD && qx >= 1 && qx <= W - 2 && qy >= 1 && qy <= H - 2) {
            const size_t q = (size_t)qy * W + qx;
            alpha_d[k] = allmap[HW + q];
            if (cn != 0.f) { nv_d[k][0] = allmap[2 * HW + q]; nv_d[k][1] = allmap[3 * HW + q]; nv_d[k][2] = allmap[4 * HW + q]; }
        }
    }
    for (int i = tid; i < HP * HP; i += 256) {
        const int r = i / HP, cc = i - r * HP;
        const int qy = y0 + r - 2, qx = x0 + cc - 2;
        float p[3] = {0.f, 0.f, 0.f};
        if (qy >= 0 && qy < H && qx >= 0 && qx < W) {
            const float sd = surf_depth_at(allmap, HW, (size_t)qy * W + qx, ratio);
            float d[3];
            ray_dir(c, qx, qy, d);
#pragma unroll
            for (int j = 0; j < 3; j++) p[j] = sd * d[j] + c.o[j];
        }
        px[i] = p[0]; py[i] = p[1]; pz[i] = p[2];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < ND; k++) {
        const int i = tid + 256 * k;
        if (i >= HD * HD) break;
        const int r = i / HD, cc = i - r * HD;
        const int qy = y0 + r - 1, qx = x0 + cc - 1;
        float ddx[3] = {0.f, 0.f, 0.f}, ddy[3] = {0.f, 0.f, 0.f}, sn[3] = {0.f, 0.f, 0.f};
        if (qx >= 1 && qx <= W - 2 && qy >= 1 && qy <= H - 2) {       // interior pixel: has a finite-difference normal
            const size_t og = (size_t)qy * W + qx;
            const int q = (r + 1) * HP + cc + 1;
            const float dxv[3] = {px[q + HP] - px[q - HP], py[q + HP] - py[q - HP], pz[q + HP] - pz[q - HP]};
            const float dyv[3] = {px[q + 1] - px[q - 1], py[q + 1] - py[q - 1], pz[q + 1] - pz[q - 1]};
            float v[3];
            cross3(dxv, dyv, v);
            const float len = sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
            const float inv = 1.f / fmaxf(len, 1e-12f);
            const float u[3] = {v[0] * inv, v[1] * inv, v[2] * inv};
            const float alpha = alpha_d[k];
            // upstream gradient w.r.t. surf_normal = u * alpha(detached)
            float g[3] = {0.f, 0.f, 0.f};
            if (gmaps) { g[0] = gmaps[6 * HW + og]; g[1] = gmaps[7 * HW + og]; g[2] = gmaps[8 * HW + og]; }
            if (cn != 0.f) {     // d/dsn of cn * (1 - rn . sn) = -cn * rn
                const float nv[3] = {nv_d[k][0], nv_d[k][1], nv_d[k][2]};
#pragma unroll
                for (int j = 0; j < 3; j++) g[j] -= cn * (c.A[3 * j] * nv[0] + c.A[3 * j + 1] * nv[1] + c.A[3 * j + 2] * nv[2]);
            }
            float gu[3] = {g[0] * alpha, g[1] * alpha, g[2] * alpha};
            float gv[3];
            if (len >= 1e-12f) {
                const float ug = u[0] * gu[0] + u[1] * gu[1] + u[2] * gu[2];
#pragma unroll
                for (int j = 0; j < 3; j++) gv[j] = (gu[j] - u[j] * ug) * inv;
            } else {
#pragma unroll
                for (int j = 0; j < 3; j++) gv[j] = gu[j] * 1e12f;      // clamp_min(eps) branch of F.normalize
            }
            cross3(dyv, gv, ddx);    // v = dx x dy:  dL/d(dx) = dy x gv,  dL/d(dy) = gv x dx
            cross3(gv, dxv, ddy);
#pragma unroll
            for (int j = 0; j < 3; j++) sn[j] = u[j] * alpha;
        }
#pragma unroll
        for (int j = 0; j < 3; j++) { dd[j][i] = ddx[j]; dd[3 + j][i] = ddy[j]; sn_s[j][i] = sn[j]; }
    }
    __syncthreads();
    if (!inside) return;
    const int q = (ly + 1) * HD + lx + 1;
    // points[y+1,x] receives +ddx of pixel (y,x) -> this pixel gathers +ddx from the row above, -ddx from the row below,
    // +ddy from the column to the left, -ddy from the column to the right (zeros where that pixel has no normal).
    float gp[3];
#pragma unroll
    for (int j = 0; j < 3; j++) gp[j] = dd[j][q - HD] - dd[j][q + HD] + dd[3 + j][q - 1] - dd[3 + j][q + 1];
    float d[3];
    ray_dir(c, gx, gy, d);
    float g_depth = gp[0] * d[0] + gp[1] * d[1] + gp[2] * d[2];
    float g_alpha = 0.f, g_dist = cd;
    float g_rn[3] = {0.f, 0.f, 0.f};
    if (gmaps) {
        g_alpha = gmaps[o];
        g_rn[0] = gmaps[HW + o]; g_rn[1] = gmaps[2 * HW + o]; g_rn[2] = gmaps[3 * HW + o];
        g_dist += gmaps[4 * HW + o];
        g_depth += gmaps[5 * HW + o];
    }
    if (cn != 0.f) {
#pragma unroll
        for (int j = 0; j < 3; j++) g_rn[j] -= cn * sn_s[j][q];
    }
    const float e = a0 / a1;
    float ga0 = 0.f, ga1 = g_alpha, ga5 = 0.f;
    if (finite_f(e) && a1 != 0.f) {
        const float ge = g_depth * (1.f - ratio);
        ga0 = ge / a1;
        ga1 -= ge * e / a1;
    }
    if (finite_f(a5)) ga5 = g_depth * ratio;
    gall[o] = ga0;
    gall[HW + o] = ga1;
#pragma unroll
    for (int i = 0; i < 3; i++) gall[(2 + i) * HW + o] = c.A[i] * g_rn[0] + c.A[3 + i] * g_rn[1] + c.A[6 + i] * g_rn[2];   // A^T g
    gall[5 * HW + o] = ga5;
    gall[6 * HW + o] = g_dist;
}


constexpr size_t post_fwd_lds() { return sizeof(float) * (3 * (PT + 2) * (PT + 2) + 8); }
constexpr size_t post_bwd_lds() { return sizeof(float) * (3 * (PT + 4) * (PT + 4) + 9 * (PT + 2) * (PT + 2)); }

}  // namespace postk
}  // namespace surfel
