// surfel_preprocess.hip — the per-surfel (HBM-streaming) kernels of the gfx950 surfel rasterizer.
//   preprocess_fwd   : per-surfel homography, AABB, SH colour, alpha>=1/255 footprint -> packed 112-B records
//   emit_instances   : (tile, surfel) pairs for every tile the footprint reaches, in depth order
//   tile_ranges      : [start,end) of every tile in the sorted instance list
//   mark_visible     : frustum test
//   preprocess_bwd   : per-surfel sum of its instance gradient records, then the chain rule into means,
//                      scales, rotations, opacity and SH
// Reference interfaces replaced: the native half of diff_surfel_rasterization (absent submodule,
// /root/reference/.gitmodules:1-3); semantics as stated in oracle/surfel_oracle.c.
// (The blend kernels live in surfel_forward.hip / surfel_backward.hip, which are compiled with
// -fno-slp-vectorize: packed-f32 VALU ops issue at ~1.6x a scalar op on gfx950 and cost extra v_movs.)
#include "surfel_common.h"
#include "surfel_kernels.h"

namespace surfel {


__device__ __constant__ float SH_C0 = 0.28209479177387814f;
__device__ __constant__ float SH_C1 = 0.4886025119029199f;
__device__ __constant__ float SH_C2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                                          -1.0925484305920792f, 0.5462742152960396f};
__device__ __constant__ float SH_C3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f,
                                          0.3731763325901154f, -0.4570457994644658f, 1.445305721320277f,
                                          -0.5900435899266435f};

// dB_k / d(x, y, z) of the real SH basis at the unit direction (x, y, z), zero above the active degree D (utils/sh_utils.py:57-112
// differentiated; the values themselves are built where they are used)
__device__ __forceinline__ void sh_basis_gradient(int D, float x, float y, float z, float (&Bx)[16], float (&By)[16], float (&Bz)[16]) {
#pragma unroll
    for (int k = 0; k < 16; k++) { Bx[k] = 0.f; By[k] = 0.f; Bz[k] = 0.f; }
    if (D > 0) {
        By[1] = -SH_C1; Bz[2] = SH_C1; Bx[3] = -SH_C1;
        if (D > 1) {
            const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
            Bx[4] = SH_C2[0] * y; By[4] = SH_C2[0] * x;
            By[5] = SH_C2[1] * z; Bz[5] = SH_C2[1] * y;
            Bx[6] = SH_C2[2] * -2.f * x; By[6] = SH_C2[2] * -2.f * y; Bz[6] = SH_C2[2] * 4.f * z;
            Bx[7] = SH_C2[3] * z; Bz[7] = SH_C2[3] * x;
            Bx[8] = SH_C2[4] * 2.f * x; By[8] = SH_C2[4] * -2.f * y;
            if (D > 2) {
                Bx[9] = SH_C3[0] * 6.f * xy; By[9] = SH_C3[0] * 3.f * (xx - yy);
                Bx[10] = SH_C3[1] * yz; By[10] = SH_C3[1] * xz; Bz[10] = SH_C3[1] * xy;
                Bx[11] = SH_C3[2] * -2.f * xy; By[11] = SH_C3[2] * (-3.f * yy + 4.f * zz - xx); Bz[11] = SH_C3[2] * 8.f * yz;
                Bx[12] = SH_C3[3] * -6.f * xz; By[12] = SH_C3[3] * -6.f * yz; Bz[12] = SH_C3[3] * 3.f * (2.f * zz - xx - yy);
                Bx[13] = SH_C3[4] * (-3.f * xx + 4.f * zz - yy); By[13] = SH_C3[4] * -2.f * xy; Bz[13] = SH_C3[4] * 8.f * xz;
                Bx[14] = SH_C3[5] * 2.f * xz; By[14] = SH_C3[5] * -2.f * yz; Bz[14] = SH_C3[5] * (xx - yy);
                Bx[15] = SH_C3[6] * 3.f * (xx - yy); By[15] = SH_C3[6] * -6.f * xy;
            }
        }
    }
}

// G[c][i] = d(SH colour c) / d(direction i) = sum_k sh[k][c] dB_k/d(i) from the surfel's 12 float4 of coefficients ([16][3] floats).
// ONE definition for preprocess_fwd (which leaves the rows behind) and for preprocess_bwd's fallback (a frame without rows): the
// backward's dL/dmeans3D is the same whichever of the two evaluated it.
__device__ __forceinline__ void sh_colour_jacobian(int D, float x, float y, float z, const float4 (&c4)[12], float (&G)[3][3]) {
    float Bx[16], By[16], Bz[16];
    sh_basis_gradient(D, x, y, z, Bx, By, Bz);
#pragma unroll
    for (int c = 0; c < 3; c++) { G[c][0] = 0.f; G[c][1] = 0.f; G[c][2] = 0.f; }
#pragma unroll
    for (int v = 0; v < 12; v++) {
        const float cv[4] = {c4[v].x, c4[v].y, c4[v].z, c4[v].w};
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const int flat = 4 * v + e, k = flat / 3, c = flat % 3;
            G[c][0] = __builtin_fmaf(Bx[k], cv[e], G[c][0]); G[c][1] = __builtin_fmaf(By[k], cv[e], G[c][1]); G[c][2] = __builtin_fmaf(Bz[k], cv[e], G[c][2]);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// preprocess_fwd: one thread per surfel.  Camera matrices are wave-uniform (scalar loads).
// HBM traffic per surfel: reads 40 B geometry (+192 B SH only when the surfel survives culling),
// writes 112 B record + 21 B bookkeeping.
// ---------------------------------------------------------------------------------------------
// sh_lds: this surfel's 12 float4 of SH coefficients in LDS (the wave's block arrived by LDS-DMA: preprocess_fwd_kernel), or NULL
__device__ __forceinline__ uint32_t preprocess_one(const PreprocessArgs& a, const int i, const float4* sh_lds = nullptr) {
    int rad_out = 0;
    uint32_t tiles = 0;
    uint32_t dkey = 0xffffffffu;      // culled surfels sort behind every visible one
    const float* __restrict__ vm = a.viewmatrix;
    const float px = a.means3D[3 * i], py = a.means3D[3 * i + 1], pz = a.means3D[3 * i + 2];
    // every per-surfel input is requested up front (one memory round trip instead of one per use site)
    const float opa_in = a.opacities[i];
    float4 q_in = make_float4(1.f, 0.f, 0.f, 0.f);
    float2 sc_in = make_float2(1.f, 1.f);
    if (a.transMat_precomp == nullptr) {
        q_in = reinterpret_cast<const float4*>(a.rotations)[i];
        sc_in = reinterpret_cast<const float2*>(a.scales)[i];
    }
    const float vx = vm[0] * px + vm[4] * py + vm[8] * pz + vm[12];
    const float vy = vm[1] * px + vm[5] * py + vm[9] * pz + vm[13];
    const float vz = vm[2] * px + vm[6] * py + vm[10] * pz + vm[14];
    do {
        if (vz <= 0.2f) break;
        float T[9];
        float nx, ny, nz;
        if (a.transMat_precomp == nullptr) {
            const float4 q = q_in;
            const float2 sc = sc_in;
            const float s = rsqrtf(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w);
            const float w = q.x * s, x = q.y * s, y = q.z * s, z = q.w * s;
            const float sx = a.scale_modifier * sc.x, sy = a.scale_modifier * sc.y;
            // columns of R (world axes of the disc): L0 = R[:,0]*sx, L1 = R[:,1]*sy, L2 = R[:,2]
            const float L0x = (1.f - 2.f * (y * y + z * z)) * sx, L0y = (2.f * (x * y + w * z)) * sx, L0z = (2.f * (x * z - w * y)) * sx;
            const float L1x = (2.f * (x * y - w * z)) * sy, L1y = (1.f - 2.f * (x * x + z * z)) * sy, L1z = (2.f * (y * z + w * x)) * sy;
            const float L2x = 2.f * (x * z + w * y), L2y = 2.f * (y * z - w * x), L2z = 1.f - 2.f * (x * x + y * y);
            // Pm = world2ndc * ndc2pix (4x3): wave-uniform, built from scalar loads of projmatrix
            float Pm[12];
            world2pix(a.projmatrix, a.W, a.H, Pm);
#pragma unroll
            for (int c = 0; c < 3; c++) {
                T[3 * c + 0] = L0x * Pm[0 * 3 + c] + L0y * Pm[1 * 3 + c] + L0z * Pm[2 * 3 + c];
                T[3 * c + 1] = L1x * Pm[0 * 3 + c] + L1y * Pm[1 * 3 + c] + L1z * Pm[2 * 3 + c];
                T[3 * c + 2] = px * Pm[0 * 3 + c] + py * Pm[1 * 3 + c] + pz * Pm[2 * 3 + c] + Pm[3 * 3 + c];
            }
            nx = vm[0] * L2x + vm[4] * L2y + vm[8] * L2z;
            ny = vm[1] * L2x + vm[5] * L2y + vm[9] * L2z;
            nz = vm[2] * L2x + vm[6] * L2y + vm[10] * L2z;
        } else {
#pragma unroll
            for (int k = 0; k < 9; k++) T[k] = a.transMat_precomp[9 * (size_t)i + k];
            nx = 0.f; ny = 0.f; nz = 1.f;
        }
        const float cosv = -(vx * nx + vy * ny + vz * nz);
        if (cosv == 0.f) break;
        const float flip = cosv > 0.f ? 1.f : -1.f;
        nx *= flip; ny *= flip; nz *= flip;

        const float t0 = CUTOFF * CUTOFF, t2 = -1.0f;
        const float dist = t0 * T[6] * T[6] + t0 * T[7] * T[7] + t2 * T[8] * T[8];
        if (dist == 0.f) break;
        const float f0 = t0 / dist, f2 = t2 / dist;
        const float cx = f0 * T[0] * T[6] + f0 * T[1] * T[7] + f2 * T[2] * T[8];
        const float cy = f0 * T[3] * T[6] + f0 * T[4] * T[7] + f2 * T[5] * T[8];
        const float hx = cx * cx - (f0 * T[0] * T[0] + f0 * T[1] * T[1] + f2 * T[2] * T[2]);
        const float hy = cy * cy - (f0 * T[3] * T[3] + f0 * T[4] * T[4] + f2 * T[5] * T[5]);
        const float ex = sqrtf(fmaxf(1e-4f, hx)), ey = sqrtf(fmaxf(1e-4f, hy));
        const float radius = ceilf(fmaxf(fmaxf(ex, ey), CUTOFF * FILTER_SIZE));
        const int irad = (int)radius;
        const Rect rc = tile_rect(cx, cy, irad, a.gx, a.gy);
        const int ntiles = (rc.x1 - rc.x0) * (rc.y1 - rc.y0);
        if (ntiles == 0) break;

        float r = 0.f, g = 0.f, b = 0.f;
        uint8_t clampbits = 0;
        if (a.colors_precomp == nullptr) {
            const float4* __restrict__ shq = reinterpret_cast<const float4*>(a.shs + (size_t)i * a.M * 3);
            float dx = px - a.campos[0], dy = py - a.campos[1], dz = pz - a.campos[2];
            const float il = rsqrtf(dx * dx + dy * dy + dz * dz);
            dx *= il; dy *= il; dz *= il;
            // basis values for the active degree (zero above it)
            float B[16];
#pragma unroll
            for (int k = 1; k < 16; k++) B[k] = 0.f;
            B[0] = SH_C0;
            int nb = 1;
            if (a.D > 0) {
                B[1] = -SH_C1 * dy; B[2] = SH_C1 * dz; B[3] = -SH_C1 * dx; nb = 4;
                if (a.D > 1) {
                    const float xx = dx * dx, yy = dy * dy, zz = dz * dz, xy = dx * dy, yz = dy * dz, xz = dx * dz;
                    B[4] = SH_C2[0] * xy; B[5] = SH_C2[1] * yz; B[6] = SH_C2[2] * (2.f * zz - xx - yy);
                    B[7] = SH_C2[3] * xz; B[8] = SH_C2[4] * (xx - yy); nb = 9;
                    if (a.D > 2) {
                        B[9] = SH_C3[0] * dy * (3.f * xx - yy); B[10] = SH_C3[1] * xy * dz;
                        B[11] = SH_C3[2] * dy * (4.f * zz - xx - yy); B[12] = SH_C3[3] * dz * (2.f * zz - 3.f * xx - 3.f * yy);
                        B[13] = SH_C3[4] * dx * (4.f * zz - xx - yy); B[14] = SH_C3[5] * dz * (xx - yy);
                        B[15] = SH_C3[6] * dx * (xx - 3.f * yy); nb = 16;
                    }
                }
            }
            // coefficients are [M][3] floats = 12 float4 for M = 16.  ALL loads of the surfel are issued back to back before any
            // use (one round trip, and each 128-B line is consumed while it is still in L2): with the loads interleaved with the
            // FMAs the compiler serialised 12 dependent round trips and the kernel fetched 2.6x its algorithmic bytes.
            // Only the float4s holding active coefficients are read (wave-uniform degree): 1 / 3 / 7 / 12.
            if (a.M == 16) {
                float4 c4[12];
#pragma unroll
                for (int v = 0; v < 12; v++) c4[v] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (sh_lds) {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the DMA is invisible to the compiler's own counting
#pragma unroll
                    for (int v = 0; v < 12; v++) c4[v] = sh_lds[v];
                } else if (a.D > 2) {
#pragma unroll
                    for (int v = 0; v < 12; v++) c4[v] = shq[v];
                } else if (a.D == 2) {
#pragma unroll
                    for (int v = 0; v < 7; v++) c4[v] = shq[v];
                } else if (a.D == 1) {
#pragma unroll
                    for (int v = 0; v < 3; v++) c4[v] = shq[v];
                } else {
                    c4[0] = shq[0];
                }
                float acc[3] = {0.f, 0.f, 0.f};
#pragma unroll
                for (int v = 0; v < 12; v++) {
                    const float cv[4] = {c4[v].x, c4[v].y, c4[v].z, c4[v].w};
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        const int flat = 4 * v + e;          // = 3*coef + channel; B is zero above the active degree
                        acc[flat % 3] += B[flat / 3] * cv[e];
                    }
                }
                r = acc[0]; g = acc[1]; b = acc[2];
                if (a.shjac) {
                    // [r6] d(colour c) / d(direction) = sum_k sh[k][c] dB_k/d(x, y, z), left behind for preprocess_bwd: the view direction's
                    // share of dL/dmeans3D then costs it 48 B per surfel instead of a second read of the 192-B SH block
                    float G[3][3];
                    sh_colour_jacobian(a.D, dx, dy, dz, c4, G);
#pragma unroll
                    for (int c = 0; c < 3; c++) {      // nine planes of P floats: every store (and the backward's load) is coalesced
#pragma unroll
                        for (int q = 0; q < 3; q++) a.shjac[(size_t)(3 * c + q) * a.P + i] = G[c][q];
                    }
                }
            } else {
                const float* __restrict__ sh = a.shs + (size_t)i * a.M * 3;
                r = SH_C0 * sh[0]; g = SH_C0 * sh[1]; b = SH_C0 * sh[2];
                if (nb > 1) {
#pragma unroll
                    for (int k = 1; k < 16; k++) {
                        if (k < nb && k < a.M) { r += B[k] * sh[3 * k]; g += B[k] * sh[3 * k + 1]; b += B[k] * sh[3 * k + 2]; }
                    }
                }
            }
            r += 0.5f; g += 0.5f; b += 0.5f;
            clampbits = (r < 0.f ? 1 : 0) | (g < 0.f ? 2 : 0) | (b < 0.f ? 4 : 0);
            r = fmaxf(r, 0.f); g = fmaxf(g, 0.f); b = fmaxf(b, 0.f);
        } else {
            r = a.colors_precomp[3 * (size_t)i]; g = a.colors_precomp[3 * (size_t)i + 1]; b = a.colors_precomp[3 * (size_t)i + 2];
        }
        // Conservative footprint of the region where this surfel can reach alpha >= 1/255:
        //   alpha = min(.99, o*exp(-rho/2)) >= 1/255  =>  rho = min(rho3d, rho2d) <= rmax = 2 ln(255 o)
        //   {rho2d <= rmax}: disc of radius sqrt(rmax/2) about (cx,cy);
        //   {rho3d <= rmax}: the projected sqrt(rmax)-sigma ellipse = the conic whose dual is M diag(rmax,rmax,-1) M^T
        //   (M = rows Tu,Tv,Tw), i.e. centre e = (D02,D12)/D22 and "covariance" S = e e^T - D[0:2,0:2]/D22.
        // S is evaluated in a frame shifted to the projected disc centre so that e e^T - D/D22 does not cancel ~1e6-sized terms, then
        // inflated (x1.002 + 0.3 px on the diagonal) for fp32; unbounded when the ellipse meets the camera plane.
        float bx0 = 1.f, bx1 = 0.f, by0 = 1.f, by1 = 0.f;       // bbox of the footprint (tile-level cull); empty
        float ecx = cx, ecy = cy, Sxx = FOOT_UNBOUNDED, Sxy = 0.f, Syy = FOOT_UNBOUNDED, r2sq = 0.f, Sdet = 1.f;
        {
            const float opa = opa_in;
            if (opa * 255.f >= 0.999f) {
                const float rmax = 2.f * __logf(fmaxf(opa * 255.f, 1.f)) * 1.0001f + 1e-3f;
                const float r2 = sqrtf(0.5f * rmax) + 0.05f;
                r2sq = r2 * r2;
                bx0 = cx - r2; bx1 = cx + r2; by0 = cy - r2; by1 = cy + r2;
                const float tw2 = T[8] * T[8];
                const float dc = rmax * (T[6] * T[6] + T[7] * T[7]) - tw2;
                bool bounded = dc < -1e-3f * tw2;
                if (bounded) {
                    const float g0 = rmax / dc, g2 = -1.f / dc;
                    // frame origin = the projected disc centre (u = v = 0), which always lies inside this ellipse.  (cx, cy) — the
                    // centre of the 3-sigma box — does not: it runs off to ~1e5 px when the 3-sigma ellipse grazes the camera plane
                    // while this smaller one is still bounded, and the shifted form then cancels catastrophically.
                    const float ox = T[2] / T[8], oy = T[5] / T[8];
                    const float U0 = T[0] - ox * T[6], U1 = T[1] - ox * T[7], U2 = T[2] - ox * T[8];
                    const float V0 = T[3] - oy * T[6], V1 = T[4] - oy * T[7], V2 = T[5] - oy * T[8];
                    const float ex_ = g0 * U0 * T[6] + g0 * U1 * T[7] + g2 * U2 * T[8];
                    const float ey_ = g0 * V0 * T[6] + g0 * V1 * T[7] + g2 * V2 * T[8];
                    const float sxx = ex_ * ex_ - (g0 * U0 * U0 + g0 * U1 * U1 + g2 * U2 * U2);
                    const float sxy = ex_ * ey_ - (g0 * U0 * V0 + g0 * U1 * V1 + g2 * U2 * V2);
                    const float syy = ey_ * ey_ - (g0 * V0 * V0 + g0 * V1 * V1 + g2 * V2 * V2);
                    constexpr float EPS = 0.09f;
                    const float ixx = fmaxf(sxx, 0.f) * 1.002f + EPS, iyy = fmaxf(syy, 0.f) * 1.002f + EPS;
                    // det(S0*1.002 + EPS*I) >= EPS*tr(S0)*1.002: the analytic floor survives the cancellation in
                    // ixx*iyy - sxy^2 for long thin diagonal footprints; the last term over-covers its rounding error
                    const float det = fmaxf(ixx * iyy - sxy * sxy, EPS * (ixx + iyy - 2.f * EPS)) + 4e-7f * ixx * iyy;
                    bounded = (ex_ == ex_) && (ey_ == ey_) && (det == det) && (ixx < 1e12f) && (iyy < 1e12f) && (fabsf(sxy) < 1e12f);
                    if (bounded) {
                        ecx = ox + ex_; ecy = oy + ey_; Sxx = ixx; Sxy = sxy; Syy = iyy; Sdet = det;
                        const float mx = sqrtf(ixx) + 1e-4f * fabsf(ecx) + 0.01f, my = sqrtf(iyy) + 1e-4f * fabsf(ecy) + 0.01f;
                        bx0 = fminf(bx0, ecx - mx); bx1 = fmaxf(bx1, ecx + mx);
                        by0 = fminf(by0, ecy - my); by1 = fmaxf(by1, ecy + my);
                    }
                }
                if (!bounded) { bx0 = by0 = -3.0e38f; bx1 = by1 = 3.0e38f; Sxx = Syy = FOOT_UNBOUNDED; Sxy = 0.f; Sdet = 1.f; }
            }
            if (!a.cull) { bx0 = by0 = -3.0e38f; bx1 = by1 = 3.0e38f; ecx = cx; ecy = cy; Sxx = Syy = FOOT_UNBOUNDED; Sxy = 0.f; Sdet = 1.f; }
        }
        float4* __restrict__ rec = reinterpret_cast<float4*>(a.rec + (size_t)i * REC_F);
        rec[0] = make_float4(T[0], T[1], T[2], T[3]);
        rec[1] = make_float4(T[4], T[5], T[6], T[7]);
        rec[2] = make_float4(T[8], cx, cy, opa_in);
        rec[3] = make_float4(nx, ny, nz, r);
        // Instances are emitted only for the tiles of the reference rect that the alpha>=1/255 bbox can
        // reach (a pure cull: skipped (tile, surfel) pairs contribute to no pixel, so images and gradients
        // are unchanged; `radii` keeps the reference definition).
        int ex0 = rc.x0, ex1 = rc.x1, ey0 = rc.y0, ey1 = rc.y1;
        {
            const float lo_x = ceilf(fmaxf(bx0, 0.f)), hi_x = floorf(fminf(bx1, (float)(a.gx * TILE)));
            const float lo_y = ceilf(fmaxf(by0, 0.f)), hi_y = floorf(fminf(by1, (float)(a.gy * TILE)));
            if (!(lo_x <= hi_x) || !(lo_y <= hi_y)) { ex1 = ex0; ey1 = ey0; }
            else {
                ex0 = max(ex0, (int)lo_x >> 4); ex1 = min(ex1, ((int)hi_x >> 4) + 1);
                ey0 = max(ey0, (int)lo_y >> 4); ey1 = min(ey1, ((int)hi_y >> 4) + 1);
                if (ex1 < ex0) ex1 = ex0;
                if (ey1 < ey0) ey1 = ey0;
            }
        }
        const uint32_t rectbits = (uint32_t)ex0 | ((uint32_t)ey0 << 10) | ((uint32_t)(ex1 - ex0) << 20);
        a.rects[i] = rectbits;
        rec[4] = make_float4(g, b, 0.f /* inst_base patched by emit_instances */, __uint_as_float(rectbits));
        rec[5] = make_float4(ecx, ecy, Sxx, Sxy);
        rec[6] = make_float4(Syy, r2sq, Sdet, 0.f);
        dkey = __float_as_uint(vz);
        a.clamped[i] = clampbits;
        rad_out = irad;
        tiles = (uint32_t)((ex1 - ex0) * (ey1 - ey0));
    } while (false);
    a.radii[i] = rad_out;
    a.tiles_touched[i] = tiles;
    a.depths[i] = vz;
    a.depth_keys[i] = tiles ? dkey : 0xffffffffu;
    // [r6] the sort value carries the surfel's tile count (packed frames): the scan over the depth order reads it from the sorted values —
    // coalesced — instead of gathering tiles_touched[id], a 128-B line per surfel (336 MB per launch at 2.1 M surfels)
    a.ident[i] = a.pack_tiles ? ((uint32_t)i | (min(tiles, PACK_TILES_MAX) << PACK_ID_BITS)) : (uint32_t)i;
    return tiles;
}

// The instance total R = sum(tiles_touched) sizes the binning buffers, so it has to reach the host.  It is summed
// here so the small D2H copy can be issued right after this kernel and the host wakes up, allocates and enqueues
// the rest of the forward WHILE the device is still depth-sorting.  One atomic per workgroup, spread over
// R_SLOTS counters (same-address device-scope atomics serialise at ~10 ns each); the host adds the slots.
#ifndef PRE_FWD_MINWG
#define PRE_FWD_MINWG 4
#endif
// DMA (large frames, launch_preprocess_fwd): the SH block is 83 % of this kernel's input bytes.  Read by its owner thread it is twelve
// 16-B loads at a 192-B stride per lane: 64 cache lines per load instruction.  A wave's 64 surfels hold their coefficients in ONE
// contiguous 12 KB span, so the wave fetches it as twelve fully coalesced 1 KB LDS-DMA instructions (global -> LDS, no VGPR, issued
// before anything else) and every lane picks its own 192 B out of LDS when it gets to the colour (full degree only: lower degrees
// read a prefix per surfel).  Same bits.  Measured (profiles/r06_ab_preprocess_dma.jsonl): 2.2 M surfels 0.220 -> 0.200 ms, 2 M random
// 0.187 -> 0.181; at 300 k surfels the 48 KB of LDS (3 workgroups per CU instead of 4) cost 1 us -> small frames keep the direct loads.
template <bool DMA>
__global__ void __launch_bounds__(256, PRE_FWD_MINWG) preprocess_fwd_kernel(PreprocessArgs a) {
    __shared__ uint32_t s_tot[4], s_vis[4];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    __shared__ float4 s_sh[DMA ? 4 : 1][DMA ? 64 * 12 : 1];      // DMA: 48 KB -> 3 workgroups per CU
    const float4* sh_lds = nullptr;
    if (DMA && a.shs != nullptr && a.colors_precomp == nullptr && a.M == 16 && a.D > 2) {
        const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
        const int first = blockIdx.x * 256 + wave * 64;
        const int npieces = 12 * min(64, a.P - first);
        const float4* __restrict__ src = reinterpret_cast<const float4*>(a.shs) + (size_t)first * 12;
        const unsigned base = __builtin_amdgcn_readfirstlane(lds_offset(&s_sh[wave][0]));      // (wave-uniform: M0 takes an SGPR)
#pragma unroll
        for (int v = 0; v < 12; v++) {
            const int p = v * 64 + lane;
            if (p < npieces) dma16(src + p, base + (unsigned)v * 1024u);
        }
        sh_lds = &s_sh[wave][lane * 12];
    }
    for (uint32_t w = (uint32_t)i; w < a.zero_a_words; w += gridDim.x * blockDim.x) a.zero_a[w] = 0u;
    for (uint32_t w = (uint32_t)i; w < a.zero_b_words; w += gridDim.x * blockDim.x) a.zero_b[w] = 0u;
    for (uint32_t w = (uint32_t)i; w < a.zero_c_words; w += gridDim.x * blockDim.x) a.zero_c[w] = 0u;
    uint32_t tiles = i < a.P ? preprocess_one(a, i, sh_lds) : 0u;
    const uint32_t vis = (uint32_t)__popcll(__ballot(tiles != 0u));     // surfels that emit at least one instance
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) tiles += __shfl_xor(tiles, o);
    if ((threadIdx.x & 63) == 0) { s_tot[threadIdx.x >> 6] = tiles; s_vis[threadIdx.x >> 6] = vis; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t t = s_tot[0] + s_tot[1] + s_tot[2] + s_tot[3];
        if (a.block_totals) a.block_totals[blockIdx.x] = t;
        if (t) {
            atomicAdd(a.total_instances + (blockIdx.x % R_SLOTS), t);
            // the blend-backward picks its walk variant from instances per emitting surfel (surfel_backward.hip)
            atomicAdd(a.total_instances + R_SLOTS + (blockIdx.x % R_SLOTS), s_vis[0] + s_vis[1] + s_vis[2] + s_vis[3]);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Binning is two-level (all integer, HBM-streaming work):
//   (1) the P surfels are radix-sorted by their float32 view-depth bits (stable, so equal depths keep
//       index order) — P-sized traffic instead of R-sized;
//   (2) emit_instances walks the surfels IN DEPTH ORDER and writes one (tile id, surfel) pair per touched
//       tile, so the instance list is already depth-ordered;
//   (3) a STABLE radix sort on the tile-id bits only (2 passes at <= 16 bits) groups instances by tile while
//       preserving depth order inside each tile.
// The result is identical to sorting 64-bit (tile << 32 | depth) keys, at ~1/5 of the bytes moved.
// emit_instances also patches the record's inst_base (first instance slot of the surfel), which the
// atomic-free backward uses to address its gradient records.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) emit_instances_kernel(int P, float* rec, const uint32_t* __restrict__ rects, const uint32_t* __restrict__ order,
                                                             uint32_t id_mask, const uint32_t* __restrict__ offsets_sorted, uint32_t* __restrict__ keys,
                                                             uint32_t* __restrict__ vals, int gx, uint32_t* __restrict__ zero_ptr,
                                                             uint32_t zero_words) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    for (uint32_t w = (uint32_t)k; w < zero_words; w += gridDim.x * blockDim.x) zero_ptr[w] = 0u;    // head of the tile sort that follows
    uint32_t off = 0, i = 0, rectbits = 0;
    int n = 0;
    if (k < P) {
        off = (k == 0) ? 0u : offsets_sorted[k - 1];
        n = (int)(offsets_sorted[k] - off);
        if (n > 0) {
            i = order[k] & id_mask;      // (packed frames: the tile count rides above the id)
            rectbits = rects[i];      // (written by preprocess_fwd for every surfel that emits)
            rec[(size_t)i * REC_F + 18] = __uint_as_float(off);
        }
    }
    // The wave writes ONE surfel's run of instances at a time (lane t writes instance t of the run): a run is contiguous in
    // keys / vals, so the stores of a wave land in one or two cache lines instead of 64 scattered words (thread-per-surfel runs:
    // 0.8 TB/s at 1.3e8 instances).  Surfel parameters come from the owning lane by v_readlane (the loop counter is uniform).
    // Worth it only for long runs (measured: 1.37 -> 0.78 ms at C5 with 13 instances per surfel, but 2.4x SLOWER at 2 - 4 per
    // surfel, where 64 mostly empty rounds per wave cost more than the scattered stores): decided per wave on its instance total.
    const int lane = threadIdx.x & 63;
    int wave_total = n;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) wave_total += __shfl_xor(wave_total, o);
    if (wave_total < 8 * 64) {
        const int x0 = rectbits & 1023, y0 = (rectbits >> 10) & 1023, w = rectbits >> 20;
        int x = 0, y = 0;
        for (int t = 0; t < n; t++) {
            keys[off + t] = (uint32_t)((y0 + y) * gx + (x0 + x));
            vals[off + t] = i;
            if (++x == w) { x = 0; ++y; }
        }
        return;
    }
    for (int s = 0; s < 64; s++) {
        const int ns = __builtin_amdgcn_readlane(n, s);
        if (ns == 0) continue;
        const uint32_t offs = (uint32_t)__builtin_amdgcn_readlane((int)off, s);
        const uint32_t is = (uint32_t)__builtin_amdgcn_readlane((int)i, s);
        const uint32_t rb = (uint32_t)__builtin_amdgcn_readlane((int)rectbits, s);
        const int x0 = rb & 1023, y0 = (rb >> 10) & 1023, w = rb >> 20;
        // row of instance t inside the rect: floor((t + 0.5) / w) evaluated in fp32 — (2t+1)/(2w) is never an integer and stays
        // >= 1/(2w) away from one, far more than the rounding error for w, rows <= 1023
        const float inv_w = 1.0f / (float)w;
        for (int t = lane; t < ns; t += 64) {
            const int y = (int)(((float)t + 0.5f) * inv_w);
            const int x = t - y * w;
            keys[offs + t] = (uint32_t)((y0 + y) * gx + (x0 + x));
            vals[offs + t] = is;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// tile_order: which tile every blend workgroup takes (BlendFwdArgs / BlendBwdArgs::tile_map).
// Uniform frames (random splats): workgroup b -> xcd_tile(b): every XCD (b % 8) walks a contiguous run of tiles, neighbours share
// surfel records in its private L2.  Object-centred / trained frames break both assumptions behind that: a few hundred tiles hold
// lists of thousands of instances while most hold none, so (a) the XCDs that own the image's middle rows do most of the work and
// (b) the heaviest tiles, started last, run alone at one wave per SIMD — measured on the trained leg: blend_bwd at 0.25 of the VALU
// issue peak against 0.47 (C2) and 0.61 (C4) on random frames (profiles/r03_trained_roofline.md).  For such frames the tiles are
// handed out longest-first and dealt round-robin over the XCDs: groups of 4 horizontally adjacent tiles (they still share records)
// are bucketed by the length class of their lists, the buckets are laid out from the longest class down, and sorted group k goes
// to XCD k % 8 — workgroup b = (i << 3) | xcd takes tile (i % 4) of sorted group (i / 4) * 8 + xcd.  Which frame is which is decided
// here, on the device, from the lists themselves.  The map only schedules: results do not depend on it.
// One workgroup; the map holds 32 * ceil(groups / 8) entries, -1 = no tile.
// ---------------------------------------------------------------------------------------------
constexpr int TO_CLASSES = 24, TO_THREADS = 1024, TO_WAVES = TO_THREADS / 64;
__device__ __forceinline__ int len_class(uint32_t len) { return len < 16u ? 0 : min(TO_CLASSES - 1, 32 - __clz(len >> 4)); }

// map_flag[0] = 1 and map[] filled: longest-first order; map_flag[0] = 0: the blend kernels use xcd_tile (no map is written).
// No same-address atomics anywhere (2 600 LDS atomics on 9 addresses made a first version of this kernel 14 us long): sums and maxima
// go through wave shuffles, the per-XCD work through a block scan of the list lengths, the placement ranks through ballots.
__global__ void __launch_bounds__(TO_THREADS) tile_order_kernel(const uint2* __restrict__ ranges, int gx, int gy, int* __restrict__ map, int map_len,
                                                                uint32_t* __restrict__ map_flag, int force, uint32_t* __restrict__ verdict) {
    __shared__ uint32_t s_wsum[TO_WAVES], s_wmax[TO_WAVES], s_bound[9], s_uniform;
    __shared__ uint32_t s_cnt[TO_CLASSES], s_off[TO_CLASSES], s_wcnt[TO_WAVES][TO_CLASSES];
    const int n = gx * gy, ggx = (gx + 3) >> 2, G = ggx * gy;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // ---- 1. instances per XCD under the contiguous order, longest list: thread t owns tiles [t k, (t + 1) k)
    const int k = (n + TO_THREADS - 1) / TO_THREADS;
    const int q = n >> 3, r = n & 7;      // xcd_tile's runs: XCD x owns q + 1 tiles if x < r, else q; run x starts at tile B(x)
    uint32_t sum = 0, mx = 0;
    for (int j = 0; j < k; j++) {
        const int t = tid * k + j;
        if (t < n) { const uint2 rg = ranges[t]; const uint32_t len = rg.y - rg.x; sum += len; mx = max(mx, len); }
    }
    uint32_t x = sum;                      // inclusive scan of the chunk sums over the workgroup
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const uint32_t y = __shfl_up(x, o); if (lane >= o) x += y; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = max(mx, (uint32_t)__shfl_xor((int)mx, o));
    if (lane == 63) s_wsum[wave] = x;
    if (lane == 0) s_wmax[wave] = mx;
    __syncthreads();
    uint32_t pre = x - sum;                // instances of the tiles ahead of this thread's chunk
    for (int w = 0; w < wave; w++) pre += s_wsum[w];
    for (int j = 0; j < k; j++) {          // a chunk that holds the first tile of an XCD's run publishes the prefix there
        const int t = tid * k + j;
        if (t < n) {
#pragma unroll
            for (int xc = 0; xc < 8; xc++) {
                const int B = xc < r ? xc * (q + 1) : r * (q + 1) + (xc - r) * q;
                if (t == B) s_bound[xc] = pre;
            }
            const uint2 rg = ranges[t];
            pre += rg.y - rg.x;
        }
    }
    if (tid == TO_THREADS - 1) s_bound[8] = pre;      // (the last thread's running prefix ends at the total)
    if (tid < TO_CLASSES) s_cnt[tid] = 0u;
    __syncthreads();
    if (tid == 0) {
        uint32_t total = 0, maxlen = 0;
        for (int w = 0; w < TO_WAVES; w++) { total += s_wsum[w]; maxlen = max(maxlen, s_wmax[w]); }
        uint32_t busiest = 0;
        for (int xc = 0; xc < 8; xc++) {
            const uint32_t lo = q == 0 && xc >= r ? total : s_bound[xc];
            const uint32_t hi = xc == 7 ? total : (q == 0 && xc + 1 >= r ? total : s_bound[xc + 1]);
            busiest = max(busiest, hi - lo);
        }
        // uniform: the busiest XCD holds at most 15 % more than its share and no list is longer than 4 average lists
        const bool uniform = total == 0u || ((unsigned long long)busiest * 800ull <= (unsigned long long)total * 115ull &&
                                             (unsigned long long)maxlen * (unsigned long long)n <= 4ull * total);
        s_uniform = force == 1 ? 1u : (force == 2 ? 0u : (uniform ? 1u : 0u));
        if (s_uniform) map_flag[0] = 0u;   // xcd_tile order
        if (verdict) verdict[0] = uniform ? 1u : 2u;      // (pinned host word: the caller skips this launch on frames of a size that keeps coming out uniform)
    }
    __syncthreads();
    if (s_uniform) return;
    // ---- 2. longest-first order over groups of 4 horizontally adjacent tiles
    for (int b = tid; b < map_len; b += TO_THREADS) map[b] = -1;
    auto group_class = [&](int g) {
        const int ty = g / ggx, tx0 = (g - ty * ggx) << 2;
        uint32_t len4 = 0;
        for (int j = 0; j < 4 && tx0 + j < gx; j++) { const uint2 rg = ranges[ty * gx + tx0 + j]; len4 += rg.y - rg.x; }
        return len_class(len4);
    };
    const unsigned long long lt = (1ull << lane) - 1ull;
    // 2a. groups per class (one LDS atomic per (wave, class present in the wave))
    for (int g0 = 0; g0 < G; g0 += TO_THREADS) {
        const int g = g0 + tid;
        const int c = g < G ? group_class(g) : -1;
        unsigned long long rest = __ballot(c >= 0);
        while (rest) {
            const int cc = __shfl(c, __builtin_ctzll(rest));
            const unsigned long long m = __ballot(c == cc);
            if (lane == __builtin_ctzll(m)) atomicAdd(&s_cnt[cc], (uint32_t)__popcll(m));
            rest &= ~m;
        }
    }
    __syncthreads();
    if (tid == 0) {
        uint32_t run = 0;
        for (int c = TO_CLASSES - 1; c >= 0; c--) { s_off[c] = run; run += s_cnt[c]; }
    }
    // 2b. placement, 1024 groups at a time: position = class offset + groups of the class in earlier waves + rank inside the wave
    // (ballots) — the order inside a class is the groups' own order, so spatial neighbours stay neighbours and the map is deterministic
    for (int g0 = 0; g0 < G; g0 += TO_THREADS) {
        for (int i = tid; i < TO_WAVES * TO_CLASSES; i += TO_THREADS) (&s_wcnt[0][0])[i] = 0u;
        __syncthreads();
        const int g = g0 + tid;
        const int c = g < G ? group_class(g) : -1;
        uint32_t rank = 0;
        unsigned long long rest = __ballot(c >= 0);
        while (rest) {
            const int cc = __shfl(c, __builtin_ctzll(rest));
            const unsigned long long m = __ballot(c == cc);
            if (c == cc) rank = (uint32_t)__popcll(m & lt);
            if (lane == __builtin_ctzll(m)) s_wcnt[wave][cc] = (uint32_t)__popcll(m);
            rest &= ~m;
        }
        __syncthreads();
        if (c >= 0) {
            uint32_t kpos = s_off[c] + rank;
            for (int w = 0; w < wave; w++) kpos += s_wcnt[w][c];
            const int ty = g / ggx, tx0 = (g - ty * ggx) << 2;
            for (int j = 0; j < 4 && tx0 + j < gx; j++) map[((((kpos >> 3) << 2) + j) << 3) | (kpos & 7u)] = ty * gx + tx0 + j;
        }
        __syncthreads();
        if (tid < TO_CLASSES) { uint32_t add = 0; for (int w = 0; w < TO_WAVES; w++) add += s_wcnt[w][tid]; s_off[tid] += add; }
        __syncthreads();
    }
    if (tid == 0) map_flag[0] = 1u;
}

// ---------------------------------------------------------------------------------------------
// tile_ranges: boundaries of each tile's run in the sorted tile-id list (ranges pre-zeroed).
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) tile_ranges_kernel(int64_t R, const uint32_t* __restrict__ keys, uint2* __restrict__ ranges) {
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= R) return;
    const uint32_t tile = keys[k];
    if (k == 0) ranges[tile].x = 0;
    else {
        const uint32_t prev = keys[k - 1];
        if (prev != tile) { ranges[prev].y = (uint32_t)k; ranges[tile].x = (uint32_t)k; }
    }
    if (k == R - 1) ranges[tile].y = (uint32_t)R;
}

__global__ void __launch_bounds__(256) mark_visible_kernel(int P, const float* __restrict__ means3D, const float* __restrict__ vm,
                                                           uint8_t* __restrict__ present) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P) return;
    const float vz = vm[2] * means3D[3 * i] + vm[6] * means3D[3 * i + 1] + vm[10] * means3D[3 * i + 2] + vm[14];
    present[i] = vz > 0.2f ? 1 : 0;
}

__device__ __constant__ float BSH_C0 = 0.28209479177387814f;
__device__ __constant__ float BSH_C1 = 0.4886025119029199f;
__device__ __constant__ float BSH_C2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                                           -1.0925484305920792f, 0.5462742152960396f};
__device__ __constant__ float BSH_C3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f,
                                           0.3731763325901154f, -0.4570457994644658f, 1.445305721320277f,
                                           -0.5900435899266435f};


constexpr int NV = 18;    // gradient values per instance record

// Does instance k (0-based) of a surfel's emitted tile rect have a gradient record?  blend_bwd writes records only for the list
// positions of a tile up to its cut (the last instance any pixel of the tile reached); lists are ordered by (depth bits, surfel
// index), so the surfel's own key decides — 8 B from a per-tile array instead of an 80-B record of zeros.
struct RecWalk {
    uint32_t key, idp, x0, rw, gx;
    uint32_t tile, col;      // tile of the current record, its column inside the rect
    __device__ __forceinline__ void init(uint32_t depth_bits, uint32_t i, uint32_t rectbits, int gx_) {
        key = depth_bits; idp = i; gx = (uint32_t)gx_;
        x0 = rectbits & 1023u; rw = rectbits >> 20;
        tile = ((rectbits >> 10) & 1023u) * gx + x0; col = 0u;
    }
    __device__ __forceinline__ bool has_record(const uint2* __restrict__ cut) const {
        const uint2 c = cut[tile];
        return key < c.x || (key == c.x && idp < c.y);
    }
    __device__ __forceinline__ void next() { col++; tile++; if (col == rw) { col = 0u; tile += gx - rw; } }
};

// ---------------------------------------------------------------------------------------------
// preprocess_bwd: one thread per surfel.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void store3(float* p, size_t i, float x, float y, float z) { p[3 * i] = x; p[3 * i + 1] = y; p[3 * i + 2] = z; }

// Every output element of every surfel is written here (zeros for culled surfels and inactive SH degrees),
// so the caller does not have to zero-fill nine gradient tensors per step.
// COOP (frames with many tile instances per surfel): the per-surfel sums of the instance records are gathered by the WAVE instead
// of by the owning thread.  A thread walking its own ~1 KB of records touches one 128-B line per 16-B load (12 % used, the lines
// bounce through L2: 3 TB/s at C5); here groups of 5 lanes read one record's five float4 as one contiguous 80 B, twelve groups
// per wave work on twelve surfels at a time and fetch the next unassigned surfel of the wave when theirs is done.  Every value is
// still added over the records in emission order, so the sums are bit-identical to the per-thread walk.
// HEAVY surfels (more than HEAVY_MIN instance records: background-sized discs of trained scenes cover thousands of tiles): their
// records are summed by the whole WAVE — lane l takes records l, l + 64, ... in order, then a butterfly over the lanes — instead
// of by one thread (or one 5-lane group) walking them one after the other: on a trained 456 k-surfel state at 1600x1060 a
// handful of such walks made preprocess_bwd 1.8 ms long (0.39 ms for 2 M random surfels with 4x the records).  The order is
// fixed, so the sums are reproducible, and both record gathers treat heavy surfels the same way (they stay bit-identical).
constexpr uint32_t HEAVY_MIN = 128;

#ifndef PRE_BWD_MINWG
#define PRE_BWD_MINWG 4
#endif
// JAC [r6]: the view direction's share of dL/dmeans3D comes from the forward's d(SH colour) / d(direction) rows (36 B per surfel,
// PreprocessArgs::shjac) instead of a second read of the 192-B SH block and 48 derivative-basis values; without them (frames of a
// forward that left none: SURFEL_OPT_NO_STREAM, M != 16) the coefficients are read again.
template <bool COOP, bool CUT, bool JAC>
__global__ void __launch_bounds__(256, PRE_BWD_MINWG) preprocess_bwd_kernel(PreprocessBwdArgs a) {
    __shared__ float4 s_sum[COOP ? 256 * 5 : 1];
    if (frame_overflowed(a.n_dev, a.n_cap)) return;      // (uniform: before any barrier)
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    float4 hs0 = make_float4(0.f, 0.f, 0.f, 0.f), hs1 = hs0, hs2 = hs0, hs3 = hs0, hs4 = hs0;
    bool heavy = false;
    {
        const int lane = threadIdx.x & 63;
        uint32_t hb = 0, hc = 0, hd = 0, hr = 0;
        if (i < a.P && a.radii[i] > 0 && (!CUT || a.has_rec[i])) {      // (CUT: a surfel no tile staged has no record at all)
            hc = a.tiles_touched[i];
            if (hc > HEAVY_MIN) {
                hb = __float_as_uint(a.rec[(size_t)i * REC_F + 18]);
                hr = __float_as_uint(a.rec[(size_t)i * REC_F + 19]);
                hd = __float_as_uint(a.depths[i]);
            }
        }
        heavy = hc > HEAVY_MIN;
        unsigned long long hv = __ballot(heavy);
        while (hv) {
            const int src = __builtin_ctzll(hv);
            hv &= hv - 1ull;
            const uint32_t b = __shfl(hb, src), c = __shfl(hc, src), key = __shfl(hd, src), rb = __shfl(hr, src);
            const uint32_t idp = (uint32_t)(blockIdx.x * blockDim.x + (threadIdx.x & ~63) + src);
            const uint32_t x0 = rb & 1023u, y0 = (rb >> 10) & 1023u, w = rb >> 20;
            const float inv_w = 1.0f / (float)w;
            float4 p0 = make_float4(0.f, 0.f, 0.f, 0.f), p1 = p0, p2 = p0, p3 = p0, p4 = p0;
            const float4* __restrict__ g4 = reinterpret_cast<const float4*>(a.grec);
            for (uint32_t r = (uint32_t)lane; r < c; r += 64u) {
                bool has = true;
                if (CUT) {
                    // row of record r inside the rect: floor((r + 0.5) / w) in fp32 (exact for w, rows <= 1023: emit_instances_kernel)
                    const uint32_t row = (uint32_t)(((float)r + 0.5f) * inv_w);
                    const uint2 ct = a.cut[(y0 + row) * (uint32_t)a.gx + x0 + (r - row * w)];
                    has = key < ct.x || (key == ct.x && idp < ct.y);
                }
                if (has) {
                    const float4* __restrict__ sp = g4 + (size_t)(b + r) * 5;
                    const float4 v0 = sp[0], v1 = sp[1], v2 = sp[2], v3 = sp[3], v4 = sp[4];
                    p0.x += v0.x; p0.y += v0.y; p0.z += v0.z; p0.w += v0.w; p1.x += v1.x; p1.y += v1.y; p1.z += v1.z; p1.w += v1.w;
                    p2.x += v2.x; p2.y += v2.y; p2.z += v2.z; p2.w += v2.w; p3.x += v3.x; p3.y += v3.y; p3.z += v3.z; p3.w += v3.w;
                    p4.x += v4.x; p4.y += v4.y;
                }
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                p0.x += __shfl_xor(p0.x, o); p0.y += __shfl_xor(p0.y, o); p0.z += __shfl_xor(p0.z, o); p0.w += __shfl_xor(p0.w, o);
                p1.x += __shfl_xor(p1.x, o); p1.y += __shfl_xor(p1.y, o); p1.z += __shfl_xor(p1.z, o); p1.w += __shfl_xor(p1.w, o);
                p2.x += __shfl_xor(p2.x, o); p2.y += __shfl_xor(p2.y, o); p2.z += __shfl_xor(p2.z, o); p2.w += __shfl_xor(p2.w, o);
                p3.x += __shfl_xor(p3.x, o); p3.y += __shfl_xor(p3.y, o); p3.z += __shfl_xor(p3.z, o); p3.w += __shfl_xor(p3.w, o);
                p4.x += __shfl_xor(p4.x, o); p4.y += __shfl_xor(p4.y, o);
            }
            if (lane == src) { hs0 = p0; hs1 = p1; hs2 = p2; hs3 = p3; hs4 = p4; }
        }
    }
    if (COOP) {
        const int lane = threadIdx.x & 63, wbase = threadIdx.x & ~63;
        uint32_t beg = 0, cnt = 0, dbits = 0, rbits = 0;
        if (i < a.P && a.radii[i] > 0 && !heavy && (!CUT || a.has_rec[i])) {
            beg = __float_as_uint(a.rec[(size_t)i * REC_F + 18]);          // q4.z: inst_base patched by emit_instances
            rbits = __float_as_uint(a.rec[(size_t)i * REC_F + 19]);        // q4.w: emitted tile rect
            cnt = a.tiles_touched[i];
            dbits = __float_as_uint(a.depths[i]);
        }
        const int grp = lane / 5, q = lane - 5 * grp;                       // lanes 60-63 idle
        const bool worker = grp < 12;
        const int leader = grp * 5;
        int owner = worker ? grp : 64;                                      // wave lane whose surfel this group is summing
        int next = 12;                                                      // next unassigned surfel (wave-uniform)
        uint32_t b = __shfl(beg, owner & 63), c = __shfl(cnt, owner & 63), r = 0;
        RecWalk rw;
        rw.init(__shfl(dbits, owner & 63), (uint32_t)(blockIdx.x * blockDim.x + wbase + (owner & 63)), __shfl(rbits, owner & 63), a.gx);
        if (!worker) c = 0;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        const float4* __restrict__ g4 = reinterpret_cast<const float4*>(a.grec);
        for (;;) {
            const bool done = worker && owner < 64 && r >= c;
            const unsigned long long req = __ballot(done && q == 0);
            if (req) {
                // groups that finished a surfel park its sums and take the next unassigned ones, in lane order
                if (done) s_sum[(wbase + owner) * 5 + q] = acc;
                const int rank = __popcll(req & ((1ull << leader) - 1ull));
                int nown = next + rank;
                next += __popcll(req);
                if (done) {
                    owner = nown < 64 ? nown : 64;
                    acc = make_float4(0.f, 0.f, 0.f, 0.f); r = 0;
                }
                const uint32_t nb = __shfl(beg, owner & 63), nc = __shfl(cnt, owner & 63);
                const uint32_t nd = __shfl(dbits, owner & 63), nr = __shfl(rbits, owner & 63);
                if (done) { b = nb; c = owner < 64 ? nc : 0u; rw.init(nd, (uint32_t)(blockIdx.x * blockDim.x + wbase + (owner & 63)), nr, a.gx); }
            }
            const bool live = worker && owner < 64 && r < c;
            if (!__any(worker && owner < 64)) break;
            if (live) {
                if (!CUT || rw.has_record(a.cut)) {      // (a record that does not exist is a record of zeros: skipping the addition keeps the bits)
                    const float4 v = g4[(size_t)(b + r) * 5 + q];
                    acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
                }
                rw.next();
                r++;
            }
        }
        __syncthreads();
    }
    if (i >= a.P) return;
    const bool precomp = a.transMat_precomp != nullptr;
    const bool vis = a.radii[i] > 0;
    float4* __restrict__ gshq = (a.shs && a.dL_dsh) ? reinterpret_cast<float4*>(a.dL_dsh + (size_t)i * a.M * 3) : nullptr;
    if (!vis) {
        a.dL_dopacity[i] = 0.f;
        if (a.dL_dnormal) store3(a.dL_dnormal, i, 0.f, 0.f, 0.f);
        if (!a.keep_colors) store3(a.dL_dcolors, i, 0.f, 0.f, 0.f);
        store3(a.dL_dmeans2D, i, 0.f, 0.f, 0.f); store3(a.dL_dmeans3D, i, 0.f, 0.f, 0.f);
        if (a.dL_dtransMat) {
#pragma unroll
            for (int q = 0; q < 9; q++) a.dL_dtransMat[9 * (size_t)i + q] = 0.f;
        }
        if (!precomp) {
            reinterpret_cast<float2*>(a.dL_dscales)[i] = make_float2(0.f, 0.f);
            reinterpret_cast<float4*>(a.dL_drots)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        if (gshq) {
            if (a.M == 16) {
#pragma unroll
                for (int v = 0; v < 12; v++) gshq[v] = make_float4(0.f, 0.f, 0.f, 0.f);
            } else {
                for (int v = 0; v < a.M * 3; v++) a.dL_dsh[(size_t)i * a.M * 3 + v] = 0.f;
            }
        }
        return;
    }
    // 1. gather-sum this surfel's instance gradient records (contiguous in emission order, fixed order)
    float g[NV];
#pragma unroll
    for (int q = 0; q < NV; q++) g[q] = 0.f;
    const float4* __restrict__ rq = reinterpret_cast<const float4*>(a.rec + (size_t)i * REC_F);
    // the whole record is read up front (one round trip; its one or two 128-B lines are consumed while still in L2)
    const float4 r4 = rq[4], r0 = rq[0], r1 = rq[1], r2 = rq[2];
    const uint32_t beg = __float_as_uint(r4.z);          // inst_base patched by emit_instances
    const uint32_t end = beg + a.tiles_touched[i];
    // two records (10 independent 16-B loads) in flight per step; the additions keep the emission order, so the sums are
    // bit-identical to the one-record walk (large frames hold ~10 records per surfel and were latency-bound here)
    auto add_rec = [&](const float4& v0, const float4& v1, const float4& v2, const float4& v3, const float4& v4) {
        g[0] += v0.x; g[1] += v0.y; g[2] += v0.z; g[3] += v0.w;
        g[4] += v1.x; g[5] += v1.y; g[6] += v1.z; g[7] += v1.w;
        g[8] += v2.x; g[9] += v2.y; g[10] += v2.z; g[11] += v2.w;
        g[12] += v3.x; g[13] += v3.y; g[14] += v3.z; g[15] += v3.w;
        g[16] += v4.x; g[17] += v4.y;
    };
    if (COOP) {
        const float4 v0 = s_sum[threadIdx.x * 5 + 0], v1 = s_sum[threadIdx.x * 5 + 1], v2 = s_sum[threadIdx.x * 5 + 2],
                     v3 = s_sum[threadIdx.x * 5 + 3], v4 = s_sum[threadIdx.x * 5 + 4];
        g[0] = v0.x; g[1] = v0.y; g[2] = v0.z; g[3] = v0.w; g[4] = v1.x; g[5] = v1.y; g[6] = v1.z; g[7] = v1.w;
        g[8] = v2.x; g[9] = v2.y; g[10] = v2.z; g[11] = v2.w; g[12] = v3.x; g[13] = v3.y; g[14] = v3.z; g[15] = v3.w;
        g[16] = v4.x; g[17] = v4.y;
    } else if (!heavy) {
    if (CUT) {
        if (a.has_rec[i]) {
        RecWalk rw;
        rw.init(__float_as_uint(a.depths[i]), (uint32_t)i, __float_as_uint(r4.w), a.gx);
        uint32_t k = beg;
        for (; k + 1 < end; k += 2) {
            const bool h0 = rw.has_record(a.cut);
            rw.next();
            const bool h1 = rw.has_record(a.cut);
            rw.next();
            const float4* __restrict__ src = reinterpret_cast<const float4*>(a.grec + (size_t)k * GREC_F);
            const float4 zz = make_float4(0.f, 0.f, 0.f, 0.f);
            float4 v0 = zz, v1 = zz, v2 = zz, v3 = zz, v4 = zz, w0 = zz, w1 = zz, w2 = zz, w3 = zz, w4 = zz;
            if (h0) { v0 = src[0]; v1 = src[1]; v2 = src[2]; v3 = src[3]; v4 = src[4]; }
            if (h1) { w0 = src[5]; w1 = src[6]; w2 = src[7]; w3 = src[8]; w4 = src[9]; }
            if (h0) add_rec(v0, v1, v2, v3, v4);
            if (h1) add_rec(w0, w1, w2, w3, w4);
        }
        if (k < end && rw.has_record(a.cut)) {
            const float4* __restrict__ src = reinterpret_cast<const float4*>(a.grec + (size_t)k * GREC_F);
            const float4 v0 = src[0], v1 = src[1], v2 = src[2], v3 = src[3], v4 = src[4];
            add_rec(v0, v1, v2, v3, v4);
        }
        }
    } else {
        uint32_t k = beg;
        for (; k + 1 < end; k += 2) {
            const float4* __restrict__ src = reinterpret_cast<const float4*>(a.grec + (size_t)k * GREC_F);
            const float4 v0 = src[0], v1 = src[1], v2 = src[2], v3 = src[3], v4 = src[4];
            const float4 w0 = src[5], w1 = src[6], w2 = src[7], w3 = src[8], w4 = src[9];
            add_rec(v0, v1, v2, v3, v4);
            add_rec(w0, w1, w2, w3, w4);
        }
        if (k < end) {
            const float4* __restrict__ src = reinterpret_cast<const float4*>(a.grec + (size_t)k * GREC_F);
            const float4 v0 = src[0], v1 = src[1], v2 = src[2], v3 = src[3], v4 = src[4];
            add_rec(v0, v1, v2, v3, v4);
        }
    }
    }
    if (heavy) {
        g[0] = hs0.x; g[1] = hs0.y; g[2] = hs0.z; g[3] = hs0.w; g[4] = hs1.x; g[5] = hs1.y; g[6] = hs1.z; g[7] = hs1.w;
        g[8] = hs2.x; g[9] = hs2.y; g[10] = hs2.z; g[11] = hs2.w; g[12] = hs3.x; g[13] = hs3.y; g[14] = hs3.z; g[15] = hs3.w;
        g[16] = hs4.x; g[17] = hs4.y;
    }
    a.dL_dopacity[i] = g[14];
    if (a.dL_dnormal) store3(a.dL_dnormal, i, g[11], g[12], g[13]);
    if (!a.keep_colors) store3(a.dL_dcolors, i, g[15], g[16], g[17]);

    const float T[9] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w, r2.x};
    float gT[9];
#pragma unroll
    for (int q = 0; q < 9; q++) gT[q] = g[q];
    const float gx2 = g[9], gy2 = g[10];
    if (gx2 != 0.f || gy2 != 0.f) {
        const float t[3] = {CUTOFF * CUTOFF, CUTOFF * CUTOFF, -1.f};
        const float d = t[0] * T[6] * T[6] + t[1] * T[7] * T[7] + t[2] * T[8] * T[8];
        const float id = 1.f / d;
        float f[3], dd = 0.f;
#pragma unroll
        for (int k = 0; k < 3; k++) {
            f[k] = t[k] * id;
            dd += (gx2 * T[k] * T[6 + k] + gy2 * T[3 + k] * T[6 + k]) * f[k];
        }
        dd *= -id;
#pragma unroll
        for (int k = 0; k < 3; k++) {
            gT[k] += gx2 * f[k] * T[6 + k];
            gT[3 + k] += gy2 * f[k] * T[6 + k];
            gT[6 + k] += gx2 * f[k] * T[k] + gy2 * f[k] * T[3 + k] + dd * t[k] * T[6 + k] * 2.f;
        }
    }
    const float px = a.means3D[3 * (size_t)i], py = a.means3D[3 * (size_t)i + 1], pz = a.means3D[3 * (size_t)i + 2];
    float dmx = 0.f, dmy = 0.f, dmz = 0.f;
    if (precomp) {
        // the centre term is folded into dL/dtransMat, and the densification statistic uses the folded value
#pragma unroll
        for (int q = 0; q < 9; q++) a.dL_dtransMat[9 * (size_t)i + q] = gT[q];
        store3(a.dL_dmeans2D, i, gT[2] * T[8] * 0.5f * (float)a.W, gT[5] * T[8] * 0.5f * (float)a.H, 0.f);
    } else {
        if (a.dL_dtransMat) {      // an intermediate on this path (the chain rule below continues to scales / rotations): optional
#pragma unroll
            for (int q = 0; q < 9; q++) a.dL_dtransMat[9 * (size_t)i + q] = g[q];
        }
        // densification statistic from the blend-stage dL/dT (before the centre term is folded in)
        store3(a.dL_dmeans2D, i, g[2] * T[8] * 0.5f * (float)a.W, g[5] * T[8] * 0.5f * (float)a.H, 0.f);
        const float* __restrict__ vm = a.viewmatrix;
        float Pm[12];
        world2pix(a.projmatrix, a.W, a.H, Pm);
        const float4 q = reinterpret_cast<const float4*>(a.rotations)[i];
        const float2 sc = reinterpret_cast<const float2*>(a.scales)[i];
        const float s = rsqrtf(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w);
        const float w = q.x * s, x = q.y * s, y = q.z * s, z = q.w * s;
        const float sx = a.scale_modifier * sc.x, sy = a.scale_modifier * sc.y;
        const float R00 = 1.f - 2.f * (y * y + z * z), R01 = 2.f * (x * y - w * z), R02 = 2.f * (x * z + w * y);
        const float R10 = 2.f * (x * y + w * z), R11 = 1.f - 2.f * (x * x + z * z), R12 = 2.f * (y * z - w * x);
        const float R20 = 2.f * (x * z - w * y), R21 = 2.f * (y * z + w * x), R22 = 1.f - 2.f * (x * x + y * y);
        // dL/dA[r][j] = sum_c gT[3c+r] * Pm[j][c]
        float dA[3][3];
#pragma unroll
        for (int r = 0; r < 3; r++)
#pragma unroll
            for (int j = 0; j < 3; j++) dA[r][j] = gT[r] * Pm[3 * j] + gT[3 + r] * Pm[3 * j + 1] + gT[6 + r] * Pm[3 * j + 2];
        // normal path
        const float nx = vm[0] * R02 + vm[4] * R12 + vm[8] * R22;
        const float ny = vm[1] * R02 + vm[5] * R12 + vm[9] * R22;
        const float nz = vm[2] * R02 + vm[6] * R12 + vm[10] * R22;
        const float vx = vm[0] * px + vm[4] * py + vm[8] * pz + vm[12];
        const float vy = vm[1] * px + vm[5] * py + vm[9] * pz + vm[13];
        const float vz = vm[2] * px + vm[6] * py + vm[10] * pz + vm[14];
        const float flip = (-(vx * nx + vy * ny + vz * nz)) > 0.f ? 1.f : -1.f;
        const float dtn0 = flip * (vm[0] * g[11] + vm[1] * g[12] + vm[2] * g[13]);
        const float dtn1 = flip * (vm[4] * g[11] + vm[5] * g[12] + vm[6] * g[13]);
        const float dtn2 = flip * (vm[8] * g[11] + vm[9] * g[12] + vm[10] * g[13]);
        // dL/dR[r][c]
        const float d00 = dA[0][0] * sx, d10 = dA[0][1] * sx, d20 = dA[0][2] * sx;
        const float d01 = dA[1][0] * sy, d11 = dA[1][1] * sy, d21 = dA[1][2] * sy;
        const float d02 = dtn0, d12 = dtn1, d22 = dtn2;
        reinterpret_cast<float2*>(a.dL_dscales)[i] = make_float2(a.scale_modifier * (dA[0][0] * R00 + dA[0][1] * R10 + dA[0][2] * R20),
                                                                 a.scale_modifier * (dA[1][0] * R01 + dA[1][1] * R11 + dA[1][2] * R21));
        dmx = dA[2][0]; dmy = dA[2][1]; dmz = dA[2][2];
        float4 gq;
        gq.x = 2.f * (x * (d21 - d12) + y * (d02 - d20) + z * (d10 - d01));
        gq.y = 2.f * (-2.f * x * (d11 + d22) + y * (d01 + d10) + z * (d02 + d20) + w * (d21 - d12));
        gq.z = 2.f * (x * (d01 + d10) - 2.f * y * (d00 + d22) + z * (d12 + d21) + w * (d02 - d20));
        gq.w = 2.f * (x * (d02 + d20) + y * (d12 + d21) - 2.f * z * (d00 + d11) + w * (d10 - d01));
        reinterpret_cast<float4*>(a.dL_drots)[i] = gq;
    }

    if (a.shs != nullptr) {
        const float dox = px - a.campos[0], doy = py - a.campos[1], doz = pz - a.campos[2];
        const float sum2 = dox * dox + doy * doy + doz * doz;
        const float il = rsqrtf(sum2);
        const float x = dox * il, y = doy * il, z = doz * il;
        const uint8_t cb = a.clamped[i];
        const float gR[3] = {(cb & 1) ? 0.f : g[15], (cb & 2) ? 0.f : g[16], (cb & 4) ? 0.f : g[17]};
        if (!a.keep_colors) store3(a.dL_dcolors, i, gR[0], gR[1], gR[2]);      // SH mode: gradient w.r.t. the pre-clamp SH colour (include/surfel_hip.h)
        float G[3][3];
        if (JAC) {
#pragma unroll
            for (int c = 0; c < 3; c++) {
#pragma unroll
                for (int q = 0; q < 3; q++) G[c][q] = a.shjac[(size_t)(3 * c + q) * a.P + i];
            }
        } else if (a.M == 16) {      // a frame whose forward left no rows: the coefficients again, through the forward's own function
            const float4* __restrict__ shq = reinterpret_cast<const float4*>(a.shs + (size_t)i * 48);
            float4 c4[12];
#pragma unroll
            for (int v = 0; v < 12; v++) c4[v] = shq[v];
            sh_colour_jacobian(a.D, x, y, z, c4, G);
        } else {
            float Bx[16], By[16], Bz[16];
            sh_basis_gradient(a.D, x, y, z, Bx, By, Bz);
            const float* __restrict__ sh = a.shs + (size_t)i * a.M * 3;
#pragma unroll
            for (int c = 0; c < 3; c++) { G[c][0] = 0.f; G[c][1] = 0.f; G[c][2] = 0.f; }
#pragma unroll
            for (int k = 0; k < 16; k++) {
                if (k < a.M) {
#pragma unroll
                    for (int c = 0; c < 3; c++) { G[c][0] += Bx[k] * sh[3 * k + c]; G[c][1] += By[k] * sh[3 * k + c]; G[c][2] += Bz[k] * sh[3 * k + c]; }
                }
            }
        }
        const float gdx = __builtin_fmaf(gR[2], G[2][0], __builtin_fmaf(gR[1], G[1][0], gR[0] * G[0][0]));
        const float gdy = __builtin_fmaf(gR[2], G[2][1], __builtin_fmaf(gR[1], G[1][1], gR[0] * G[0][1]));
        const float gdz = __builtin_fmaf(gR[2], G[2][2], __builtin_fmaf(gR[1], G[1][2], gR[0] * G[0][2]));
        if (gshq != nullptr || (a.dL_dsh != nullptr && a.M != 16)) {
            // dL/dsh = B (x) gR for the active degree (zero above it)
            float B[16];
#pragma unroll
            for (int k = 0; k < 16; k++) B[k] = 0.f;
            B[0] = BSH_C0;
            if (a.D > 0) {
                B[1] = -BSH_C1 * y; B[2] = BSH_C1 * z; B[3] = -BSH_C1 * x;
                if (a.D > 1) {
                    const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
                    B[4] = BSH_C2[0] * xy; B[5] = BSH_C2[1] * yz; B[6] = BSH_C2[2] * (2.f * zz - xx - yy); B[7] = BSH_C2[3] * xz; B[8] = BSH_C2[4] * (xx - yy);
                    if (a.D > 2) {
                        B[9] = BSH_C3[0] * y * (3.f * xx - yy); B[10] = BSH_C3[1] * xy * z; B[11] = BSH_C3[2] * y * (4.f * zz - xx - yy);
                        B[12] = BSH_C3[3] * z * (2.f * zz - 3.f * xx - 3.f * yy); B[13] = BSH_C3[4] * x * (4.f * zz - xx - yy);
                        B[14] = BSH_C3[5] * z * (xx - yy); B[15] = BSH_C3[6] * x * (xx - 3.f * yy);
                    }
                }
            }
            if (a.M == 16) {
#pragma unroll
                for (int v = 0; v < 12; v++) {
                    float o[4];
#pragma unroll
                    for (int e = 0; e < 4; e++) { const int flat = 4 * v + e; o[e] = B[flat / 3] * gR[flat % 3]; }
                    gshq[v] = make_float4(o[0], o[1], o[2], o[3]);
                }
            } else {
                float* __restrict__ gsh = a.dL_dsh + (size_t)i * a.M * 3;
#pragma unroll
                for (int k = 0; k < 16; k++) {
                    if (k < a.M) { gsh[3 * k] = B[k] * gR[0]; gsh[3 * k + 1] = B[k] * gR[1]; gsh[3 * k + 2] = B[k] * gR[2]; }
                }
                for (int k = 16; k < a.M; k++) { gsh[3 * k] = 0.f; gsh[3 * k + 1] = 0.f; gsh[3 * k + 2] = 0.f; }
            }
        }
        const float il3 = il * il * il;
        dmx += ((sum2 - dox * dox) * gdx - doy * dox * gdy - doz * dox * gdz) * il3;
        dmy += (-dox * doy * gdx + (sum2 - doy * doy) * gdy - doz * doy * gdz) * il3;
        dmz += (-dox * doz * gdx - doy * doz * gdy + (sum2 - doz * doz) * gdz) * il3;
    }
    store3(a.dL_dmeans3D, i, dmx, dmy, dmz);
}

// ------------------------------------------------------------------------------- launchers
void launch_preprocess_fwd(const PreprocessArgs& a, hipStream_t s) {
    if (a.P <= 0) return;
    if (a.P >= (1 << 19)) hipLaunchKernelGGL(preprocess_fwd_kernel<true>, dim3((a.P + 255) / 256), dim3(256), 0, s, a);
    else hipLaunchKernelGGL(preprocess_fwd_kernel<false>, dim3((a.P + 255) / 256), dim3(256), 0, s, a);
}
void launch_emit_instances(int P, float* rec, const uint32_t* rects, const uint32_t* order, uint32_t id_mask, const uint32_t* offsets_sorted, uint32_t* keys,
                           uint32_t* vals, int gx, uint32_t* zero_ptr, uint32_t zero_words, hipStream_t s) {
    if (P > 0) hipLaunchKernelGGL(emit_instances_kernel, dim3((P + 255) / 256), dim3(256), 0, s, P, rec, rects, order, id_mask, offsets_sorted, keys, vals,
                                  gx, zero_ptr, zero_words);
}
// capacity binning: the count is on the device, the grid covers the capacity
__global__ void __launch_bounds__(256) tile_ranges_devn_kernel(const uint32_t* __restrict__ n_dev, uint32_t cap, const uint32_t* __restrict__ keys,
                                                               uint2* __restrict__ ranges) {
    const uint32_t R = min(*n_dev, cap);
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= R) return;
    const uint32_t tile = keys[k];
    if (k == 0) ranges[tile].x = 0;
    else {
        const uint32_t prev = keys[k - 1];
        if (prev != tile) { ranges[prev].y = k; ranges[tile].x = k; }
    }
    if (k == R - 1) ranges[tile].y = R;
}
void launch_tile_ranges_devn(size_t cap, const uint32_t* n_dev, const uint32_t* keys, uint2* ranges, hipStream_t s) {
    if (cap > 0) hipLaunchKernelGGL(tile_ranges_devn_kernel, dim3((unsigned)((cap + 255) / 256)), dim3(256), 0, s, n_dev, (uint32_t)cap, keys, ranges);
}
int tile_map_len(int gx, int gy) { return 32 * ((((gx + 3) >> 2) * gy + 7) / 8); }
void launch_tile_order(const uint2* ranges, int gx, int gy, int* map, uint32_t* map_flag, int force, uint32_t* verdict, hipStream_t s) {
    hipLaunchKernelGGL(tile_order_kernel, dim3(1), dim3(TO_THREADS), 0, s, ranges, gx, gy, map, tile_map_len(gx, gy), map_flag, force, verdict);
}
void launch_tile_ranges(int64_t R, const uint32_t* keys, uint2* ranges, hipStream_t s) {
    if (R > 0) hipLaunchKernelGGL(tile_ranges_kernel, dim3((unsigned)((R + 255) / 256)), dim3(256), 0, s, R, keys, ranges);
}
void launch_mark_visible(int P, const float* means3D, const float* vm, uint8_t* present, hipStream_t s) {
    if (P > 0) hipLaunchKernelGGL(mark_visible_kernel, dim3((P + 255) / 256), dim3(256), 0, s, P, means3D, vm, present);
}
// dL/dcolour of every surfel = sum over its gradient records (emission order, one record at a time: the additions of
// preprocess_bwd's gather in the same order) of floats 15-17, clamp-masked in SH mode.  12 of the 80 record bytes are used, but the records of a surfel are
// contiguous, so the lines fetched here are the ones preprocess_bwd reads next.
__global__ void __launch_bounds__(256) colour_gradients_kernel(PreprocessBwdArgs a) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.P || frame_overflowed(a.n_dev, a.n_cap)) return;
    float c0 = 0.f, c1 = 0.f, c2 = 0.f;
    if (a.radii[i] > 0) {
        const uint32_t beg = __float_as_uint(a.rec[(size_t)i * REC_F + 18]);
        const uint32_t end = beg + a.tiles_touched[i];
        RecWalk rw;
        bool end_skip = false;
        rw.init(__float_as_uint(a.depths[i]), (uint32_t)i, __float_as_uint(a.rec[(size_t)i * REC_F + 19]), a.gx);
        if (a.cut && !a.has_rec[i]) end_skip = true;
        for (uint32_t k = beg; k < end && !end_skip; k++, rw.next()) {
            if (a.cut && !rw.has_record(a.cut)) continue;
            const float* __restrict__ src = a.grec + (size_t)k * GREC_F;
            const float v0 = src[15];
            const float2 v1 = *reinterpret_cast<const float2*>(src + 16);
            c0 += v0; c1 += v1.x; c2 += v1.y;
        }
        if (a.shs != nullptr) {      // SH mode: gradient w.r.t. the pre-clamp SH colour, as preprocess_bwd stores it
            const uint8_t cb = a.clamped[i];
            c0 = (cb & 1) ? 0.f : c0; c1 = (cb & 2) ? 0.f : c1; c2 = (cb & 4) ? 0.f : c2;
        }
    }
    store3(a.dL_dcolors, i, c0, c1, c2);
}

void launch_colour_gradients(const PreprocessBwdArgs& a, hipStream_t s) {
    if (a.P > 0) hipLaunchKernelGGL(colour_gradients_kernel, dim3((a.P + 255) / 256), dim3(256), 0, s, a);
}

void launch_preprocess_bwd(const PreprocessBwdArgs& a, hipStream_t s) {
    if (a.P <= 0) return;
    const dim3 grid((a.P + 255) / 256), block(256);
    const bool jac = a.shjac != nullptr && a.shs != nullptr && a.M == 16;
    if (a.coop) {
        if (a.cut) { if (jac) hipLaunchKernelGGL((preprocess_bwd_kernel<true, true, true>), grid, block, 0, s, a); else hipLaunchKernelGGL((preprocess_bwd_kernel<true, true, false>), grid, block, 0, s, a); }
        else { if (jac) hipLaunchKernelGGL((preprocess_bwd_kernel<true, false, true>), grid, block, 0, s, a); else hipLaunchKernelGGL((preprocess_bwd_kernel<true, false, false>), grid, block, 0, s, a); }
    } else {
        if (a.cut) { if (jac) hipLaunchKernelGGL((preprocess_bwd_kernel<false, true, true>), grid, block, 0, s, a); else hipLaunchKernelGGL((preprocess_bwd_kernel<false, true, false>), grid, block, 0, s, a); }
        else { if (jac) hipLaunchKernelGGL((preprocess_bwd_kernel<false, false, true>), grid, block, 0, s, a); else hipLaunchKernelGGL((preprocess_bwd_kernel<false, false, false>), grid, block, 0, s, a); }
    }
}

}  // namespace surfel
