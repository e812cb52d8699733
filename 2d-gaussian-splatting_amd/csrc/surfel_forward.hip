// surfel_forward.hip — forward kernels of the gfx950 surfel rasterizer.
//   preprocess_fwd   : per-surfel homography, AABB, SH colour  -> packed 80-B records
//   emit_instances   : (tile, depth) keys for every touched tile
//   tile_ranges      : [start,end) of every tile in the sorted instance list
//   blend_fwd        : per-tile front-to-back alpha blend, 10 output channels
// Reference interfaces replaced: the native half of diff_surfel_rasterization (absent submodule,
// /root/reference/.gitmodules:1-3); semantics as stated in oracle/surfel_oracle.c.
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

// ---------------------------------------------------------------------------------------------
// preprocess_fwd: one thread per surfel.  Camera matrices are wave-uniform (scalar loads).
// HBM traffic per surfel: reads 40 B geometry (+192 B SH only when the surfel survives culling),
// writes 112 B record + 21 B bookkeeping.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) preprocess_fwd_kernel(PreprocessArgs a) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.P) return;
    int rad_out = 0;
    uint32_t tiles = 0;
    uint32_t dkey = 0xffffffffu;      // culled surfels sort behind every visible one
    const float* __restrict__ vm = a.viewmatrix;
    const float px = a.means3D[3 * i], py = a.means3D[3 * i + 1], pz = a.means3D[3 * i + 2];
    const float vx = vm[0] * px + vm[4] * py + vm[8] * pz + vm[12];
    const float vy = vm[1] * px + vm[5] * py + vm[9] * pz + vm[13];
    const float vz = vm[2] * px + vm[6] * py + vm[10] * pz + vm[14];
    do {
        if (vz <= 0.2f) break;
        float T[9];
        float nx, ny, nz;
        if (a.transMat_precomp == nullptr) {
            const float4 q = reinterpret_cast<const float4*>(a.rotations)[i];
            const float2 sc = reinterpret_cast<const float2*>(a.scales)[i];
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
            // basis values for the active degree
            float B[16];
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
            // coefficients are [M][3] floats = 12 float4 for M = 16; walk them as float4 (16-B loads),
            // statically indexed (a runtime-indexed B[] would live in scratch), skipping the float4s
            // that hold only inactive coefficients (wave-uniform branch)
            if (a.M == 16) {
                float acc[3] = {0.f, 0.f, 0.f};
#pragma unroll
                for (int v = 0; v < 12; v++) {
                    if (4 * v < 3 * nb) {
                        const float4 c4 = shq[v];
                        const float cv[4] = {c4.x, c4.y, c4.z, c4.w};
#pragma unroll
                        for (int e = 0; e < 4; e++) {
                            const int flat = 4 * v + e;          // = 3*coef + channel
                            if (flat / 3 < 16) acc[flat % 3] += (flat / 3 < nb ? B[flat / 3] : 0.f) * cv[e];
                        }
                    }
                }
                r = acc[0]; g = acc[1]; b = acc[2];
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
        // S is evaluated in a frame shifted to (cx,cy) so that e e^T - D/D22 does not cancel ~1e6-sized terms, then
        // inflated (x1.002 + 0.3 px on the diagonal) for fp32; unbounded when the ellipse meets the camera plane.
        float bx0 = 1.f, bx1 = 0.f, by0 = 1.f, by1 = 0.f;       // bbox of the footprint (tile-level cull); empty
        float ecx = cx, ecy = cy, Sxx = FOOT_UNBOUNDED, Sxy = 0.f, Syy = FOOT_UNBOUNDED, r2sq = 0.f, Sdet = 1.f;
        {
            const float opa = a.opacities[i];
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
                    const float U0 = T[0] - cx * T[6], U1 = T[1] - cx * T[7], U2 = T[2] - cx * T[8];
                    const float V0 = T[3] - cy * T[6], V1 = T[4] - cy * T[7], V2 = T[5] - cy * T[8];
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
                        ecx = cx + ex_; ecy = cy + ey_; Sxx = ixx; Sxy = sxy; Syy = iyy; Sdet = det;
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
        rec[2] = make_float4(T[8], cx, cy, a.opacities[i]);
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
    a.ident[i] = (uint32_t)i;
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
__global__ void __launch_bounds__(256) emit_instances_kernel(int P, float* rec, const uint32_t* __restrict__ order,
                                                             const uint32_t* __restrict__ offsets_sorted, uint32_t* __restrict__ keys,
                                                             uint32_t* __restrict__ vals, int gx) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= P) return;
    const uint32_t off = (k == 0) ? 0u : offsets_sorted[k - 1];
    const int n = (int)(offsets_sorted[k] - off);
    if (n == 0) return;
    const uint32_t i = order[k];
    const uint32_t rectbits = __float_as_uint(rec[(size_t)i * REC_F + 19]);
    const int x0 = rectbits & 1023, y0 = (rectbits >> 10) & 1023, w = rectbits >> 20;
    rec[(size_t)i * REC_F + 18] = __uint_as_float(off);
    int x = 0, y = 0;
    for (int t = 0; t < n; t++) {
        keys[off + t] = (uint32_t)((y0 + y) * gx + (x0 + x));
        vals[off + t] = i;
        if (++x == w) { x = 0; ++y; }
    }
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

// ---------------------------------------------------------------------------------------------
// blend_fwd: one workgroup (4 waves) per 16x16 tile.  Surfel records of the tile's sorted list are staged through
// LDS 256 at a time (20 KB, one 80-B gather per thread).  Each DPP row of 16 lanes owns a 4x4-pixel sub-tile and
// walks its own bitmask of the staged instances whose alpha>=1/255 bbox reaches that sub-tile: a wave therefore
// blends FOUR different instances per pass of the loop body, and a small surfel occupies issue slots only where
// it can contribute.  Per-pixel order is the staged (depth) order, so results equal the whole-tile walk.
// ---------------------------------------------------------------------------------------------
constexpr int MW = 8;             // 32-bit mask words per staged batch of 256
constexpr int MSTRIDE = MW + 2;   // + zero sentinel word, padded so each sub-tile's words start 8-B aligned

__global__ void __launch_bounds__(BLOCK) blend_fwd_kernel(BlendFwdArgs a) {
    __shared__ float4 s_rec[BLOCK * 5];
    __shared__ __attribute__((aligned(8))) uint32_t s_mask[16 * MSTRIDE];    // [sub-tile][word]
    const int tile = xcd_tile(blockIdx.x, a.gx * a.gy);
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
            // ray-splat intersection, branch-free
            const float Twx = q1.z, Twy = q1.w, Twz = q2.x;
            const float kx = pxf * Twx - q0.x, ky = pxf * Twy - q0.y, kz = pxf * Twz - q0.z;
            const float lx_ = pyf * Twx - q0.w, ly_ = pyf * Twy - q1.x, lz_ = pyf * Twz - q1.y;
            const float p0 = ky * lz_ - kz * ly_, p1 = kz * lx_ - kx * lz_, p2 = kx * ly_ - ky * lx_;
            const float ip = __builtin_amdgcn_rcpf(p2);
            const float sx = p0 * ip, sy = p1 * ip;
            const float rho3d = sx * sx + sy * sy;
            const float dx = q2.y - pxf, dy = q2.z - pyf;
            const float rho2d = FILTER_INV_SQUARE * (dx * dx + dy * dy);
            const float depth = (rho3d <= rho2d) ? (sx * Twx + sy * Twy) + Twz : Twz;
            const float alpha = fminf(ALPHA_MAX, q2.w * __expf(-0.5f * fminf(rho3d, rho2d)));
            const bool ok = act & (!done) & (p2 != 0.f) & (depth >= NEAR_N) & (alpha >= ALPHA_MIN);
            if (__any(ok)) {
                if (ok) {
                    const float testT = T * (1.f - alpha);
                    if (testT < T_EPS) done = true;       // the terminating surfel is not composited
                    else {
                        const uint32_t contributor = base + j + 1;
                        const float4 q3 = s_rec[j * 5 + 3], q4 = s_rec[j * 5 + 4];
                        const float w = alpha * T;
                        const float mm = MC1 - (MC1 * NEAR_N) * __builtin_amdgcn_rcpf(depth);
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

__global__ void __launch_bounds__(256) mark_visible_kernel(int P, const float* __restrict__ means3D, const float* __restrict__ vm,
                                                           uint8_t* __restrict__ present) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P) return;
    const float vz = vm[2] * means3D[3 * i] + vm[6] * means3D[3 * i + 1] + vm[10] * means3D[3 * i + 2] + vm[14];
    present[i] = vz > 0.2f ? 1 : 0;
}

// ------------------------------------------------------------------------------- launchers
void launch_preprocess_fwd(const PreprocessArgs& a, hipStream_t s) {
    if (a.P > 0) hipLaunchKernelGGL(preprocess_fwd_kernel, dim3((a.P + 255) / 256), dim3(256), 0, s, a);
}
void launch_emit_instances(int P, float* rec, const uint32_t* order, const uint32_t* offsets_sorted, uint32_t* keys, uint32_t* vals,
                           int gx, hipStream_t s) {
    if (P > 0) hipLaunchKernelGGL(emit_instances_kernel, dim3((P + 255) / 256), dim3(256), 0, s, P, rec, order, offsets_sorted, keys, vals, gx);
}
void launch_tile_ranges(int64_t R, const uint32_t* keys, uint2* ranges, hipStream_t s) {
    if (R > 0) hipLaunchKernelGGL(tile_ranges_kernel, dim3((unsigned)((R + 255) / 256)), dim3(256), 0, s, R, keys, ranges);
}
void launch_blend_fwd(const BlendFwdArgs& a, hipStream_t s) {
    hipLaunchKernelGGL(blend_fwd_kernel, dim3(a.gx * a.gy), dim3(BLOCK), 0, s, a);
}
void launch_mark_visible(int P, const float* means3D, const float* vm, uint8_t* present, hipStream_t s) {
    if (P > 0) hipLaunchKernelGGL(mark_visible_kernel, dim3((P + 255) / 256), dim3(256), 0, s, P, means3D, vm, present);
}

}  // namespace surfel
