/*
 * surfel_oracle.c — CPU restatement of the 2DGS surfel rasterizer (forward + backward).
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path (2d-gaussian-splatting_amd/) may
 * import, link or call this file; only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg use it, and only as the checker / the reported CPU baseline.
 *
 * PARITY STATUS: **parity unpinned**.  The algorithm lives in the un-vendored, un-pinned
 * submodule hbb1/diff-surfel-rasterization (/root/reference/.gitmodules:1-3, directory empty).
 * The reference holds no tests / golden vectors for this path.  What pins this oracle:
 *   (1) the in-tree Python twin of the preprocess homography
 *       /root/reference/gaussian_renderer/__init__.py:64-75 +
 *       /root/reference/scene/gaussian_model.py:27-33 +
 *       /root/reference/utils/general_utils.py:78-110          -> tests/golden/ref_intree_*.npz
 *   (2) the in-tree SH evaluation /root/reference/utils/sh_utils.py:57-112 (+0.5, clamp_min 0 at
 *       gaussian_renderer/__init__.py:90-91)
 *   (3) the allmap channel contract /root/reference/gaussian_renderer/__init__.py:118-135
 *   (4) the paper (arXiv 2403.17888) eqs. 8-13, 17 and the appendix' O(N) distortion form
 *   (5) an independent dense fp64 PyTorch/autograd statement (oracle/dense_autograd.py) that
 *       checks the hand-derived backward below is the gradient of the forward.
 * Everything else (constants, thresholds, sort key, tile rect) is restated from the published
 * upstream algorithm as recalled and is marked [UPSTREAM-RECALL].
 *
 * Build: see oracle/Makefile.  Compiled twice: real=double (checker) and real=float
 * (-DORACLE_F32, the timed "port" CPU baseline).  OpenMP over splats / tiles.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef ORACLE_MCA
/* Monte Carlo arithmetic build (tests only; compiled as C++ by oracle/Makefile -> liboracle_mca.so): `real` carries a double whose
 * value is multiplied by (1 + 2^-24 r), r uniform in [-1, 1), after EVERY arithmetic operation — the rounding error of an fp32
 * evaluation (round to nearest: relative error <= 2^-24), drawn at random instead of determined by the operand bits.  One run
 * is one plausible fp32 evaluation of this very algorithm (decisions included); N runs with different seeds show which output
 * elements fp32 arithmetic DETERMINES to a given tolerance and which it does not (a decision sitting on its threshold, a
 * cancelling sum).  Parker, "Monte Carlo arithmetic", 1997.  The draws are a function of (seed, work item): reproducible. */
#include <cmath>
static uint64_t g_mca_seed = 1;
static thread_local uint64_t t_mca_state = 0x9E3779B97F4A7C15ull;
static inline void mca_reseed(uint64_t item) {
    uint64_t z = (g_mca_seed * 0xD1342543DE82EF95ull) ^ (item + 0x9E3779B97F4A7C15ull) * 0xBF58476D1CE4E5B9ull;
    z ^= z >> 31; z *= 0x94D049BB133111EBull; z ^= z >> 29;
    t_mca_state = z | 1ull;
}
static inline double mca_round(double x) {
    uint64_t s = t_mca_state;
    s ^= s << 13; s ^= s >> 7; s ^= s << 17;
    t_mca_state = s;
    const double r = (double)(int64_t)(s >> 11) * (1.0 / 4503599627370496.0) - 1.0;      /* [-1, 1) */
    return x * (1.0 + 5.9604644775390625e-8 * r);
}
struct real {
    double v;
    real() {}
    real(double x) : v(x) {}
    explicit operator double() const { return v; }
    explicit operator float() const { return (float)v; }
    explicit operator int() const { return (int)v; }
};
static inline real operator+(real a, real b) { return real(mca_round(a.v + b.v)); }
static inline real operator-(real a, real b) { return real(mca_round(a.v - b.v)); }
static inline real operator*(real a, real b) { return real(mca_round(a.v * b.v)); }
static inline real operator/(real a, real b) { return real(mca_round(a.v / b.v)); }
static inline real operator-(real a) { return real(-a.v); }
static inline real& operator+=(real& a, real b) { a = a + b; return a; }
static inline real& operator-=(real& a, real b) { a = a - b; return a; }
static inline real& operator*=(real& a, real b) { a = a * b; return a; }
static inline bool operator<(real a, real b) { return a.v < b.v; }
static inline bool operator<=(real a, real b) { return a.v <= b.v; }
static inline bool operator>(real a, real b) { return a.v > b.v; }
static inline bool operator>=(real a, real b) { return a.v >= b.v; }
static inline bool operator==(real a, real b) { return a.v == b.v; }
static inline bool operator!=(real a, real b) { return a.v != b.v; }
static inline real R_EXP(real a) { return real(mca_round(mca_round(std::exp(a.v)))); }      /* (hardware exp: one more ulp) */
static inline real R_SQRT(real a) { return real(mca_round(std::sqrt(a.v))); }
static inline real R_CEIL(real a) { return real(std::ceil(a.v)); }
#define MCA_RESEED(item) mca_reseed((uint64_t)(item))
#define ATOMIC_ADD(dst, val) do { const double t_ = (val).v; double* p_ = &(dst).v; _Pragma("omp atomic") *p_ += t_; } while (0)
#elif defined(ORACLE_F32)
typedef float real;
#define R_EXP expf
#define R_SQRT sqrtf
#define R_CEIL ceilf
#else
typedef double real;
#define R_EXP exp
#define R_SQRT sqrt
#define R_CEIL ceil
#endif
#ifndef ORACLE_MCA
#define MCA_RESEED(item) ((void)0)
#define ATOMIC_ADD(dst, val) do { _Pragma("omp atomic") (dst) += (val); } while (0)
#endif
#ifdef __cplusplus
extern "C" {
#endif

/* [UPSTREAM-RECALL] constants of the rasterizer */
#define TILE 16
#define NEAR_N ((real)0.2)
#define FAR_N ((real)100.0)
#define FILTER_SIZE ((real)0.707106)     /* sqrt(2)/2 low-pass sigma (paper eq. 11) */
#define FILTER_INV_SQUARE ((real)2.0)
#define CUTOFF ((real)3.0)
#define ALPHA_MIN ((real)(1.0 / 255.0))
#define ALPHA_MAX ((real)0.99)
#define T_EPS ((real)0.0001)

/* SH constants: /root/reference/utils/sh_utils.py:26-44 */
static const real SH_C0 = (real)0.28209479177387814;
static const real SH_C1 = (real)0.4886025119029199;
static const real SH_C2[5] = {(real)1.0925484305920792, (real)-1.0925484305920792, (real)0.31539156525252005,
                              (real)-1.0925484305920792, (real)0.5462742152960396};
static const real SH_C3[7] = {(real)-0.5900435899266435, (real)2.890611442640554, (real)-0.4570457994644658,
                              (real)0.3731763325901154, (real)-0.4570457994644658, (real)1.445305721320277,
                              (real)-0.5900435899266435};

typedef struct {
    int P;              /* number of surfels */
    int D;              /* active SH degree 0..3 */
    int M;              /* SH coefficients stored per surfel (16) */
    int W, H;
    double tan_fovx, tan_fovy;   /* unused by the maths (pinhole centred, README.md:173) */
    double scale_modifier;
} oracle_params;

int oracle_real_size(void) { return (int)sizeof(real); }
#ifdef ORACLE_MCA
void oracle_set_mca_seed(uint64_t seed) { g_mca_seed = seed ? seed : 1; }
#endif

/* Decision signatures (tests only).  With buffers registered, the forward blend leaves one 64-bit word per pixel and per surfel
 * that identifies the DECISIONS taken there: which (pixel, surfel) pairs were composited, on which branch (rho3d <= rho2d) each
 * one ran, which surfel carries the pixel's median depth.  Two runs whose words agree took the same decisions; a word that
 * differs between the fp64 run and a Monte-Carlo-arithmetic run marks a threshold crossing ("threshold-crossing
 * pixels exempt, count reported"). */
static uint64_t* g_sig_pix = NULL;
static uint64_t* g_sig_surf = NULL;
static double* g_extent = NULL;      /* [P]: the AABB half-extent BEFORE ceil() (0 for culled surfels): how far the radius is from flipping */
void oracle_set_signature_buffers(uint64_t* pix, uint64_t* surf, double* extent) { g_sig_pix = pix; g_sig_surf = surf; g_extent = extent; }
static inline uint64_t sig_mix(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull; x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull; x = (x ^ (x >> 27)) * 0x94D049BB133111EBull; return x ^ (x >> 31);
}

static inline int imin(int a, int b) { return a < b ? a : b; }
static inline int imax(int a, int b) { return a > b ? a : b; }

/* Tile rectangle touched by a disc of integer radius r centred at (px,py).
 * [UPSTREAM-RECALL] float division then truncation, clamped to the tile grid. */
#ifdef ORACLE_MCA
#define RAW(x) ((x).v)      /* (no noise here: preprocess and binning must derive the SAME rectangle from the same centre and radius) */
#else
#define RAW(x) (x)
#endif
static void tile_rect(real px_, real py_, int r, int gx, int gy, int* x0, int* y0, int* x1, int* y1) {
    const
#ifdef ORACLE_F32
        float
#else
        double
#endif
        px = RAW(px_), py = RAW(py_);
    *x0 = imin(gx, imax(0, (int)((px - r) / TILE)));
    *y0 = imin(gy, imax(0, (int)((py - r) / TILE)));
    *x1 = imin(gx, imax(0, (int)((px + r + TILE - 1) / TILE)));
    *y1 = imin(gy, imax(0, (int)((py + r + TILE - 1) / TILE)));
}

/* Unit-quaternion (w,x,y,z — /root/reference/utils/general_utils.py:78-101) to rotation matrix,
 * row-major R[r][c]; the quaternion is re-normalised here as upstream does. */
static void quat_to_R(const real q[4], real R[3][3]) {
    real s = 1 / R_SQRT(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    real w = q[0] * s, x = q[1] * s, y = q[2] * s, z = q[3] * s;
    R[0][0] = 1 - 2 * (y * y + z * z); R[0][1] = 2 * (x * y - w * z);     R[0][2] = 2 * (x * z + w * y);
    R[1][0] = 2 * (x * y + w * z);     R[1][1] = 1 - 2 * (x * x + z * z); R[1][2] = 2 * (y * z - w * x);
    R[2][0] = 2 * (x * z - w * y);     R[2][1] = 2 * (y * z + w * x);     R[2][2] = 1 - 2 * (x * x + y * y);
}

/* P = world2ndc * ndc2pix (4x3), with world2ndc = full_proj_transform as torch stores it
 * (row-vector convention, /root/reference/scene/cameras.py:58) and the (W-1)/2 pixel-centre
 * offset of /root/reference/gaussian_renderer/__init__.py:69-74 (columns x, y, w). */
static void world2pix(const float* pm, int W, int H, real Pm[4][3]) {
    for (int r = 0; r < 4; r++) {
        real m0 = pm[4 * r + 0], m1 = pm[4 * r + 1], m3 = pm[4 * r + 3];
        Pm[r][0] = m0 * ((real)W / 2) + m3 * ((real)(W - 1) / 2);
        Pm[r][1] = m1 * ((real)H / 2) + m3 * ((real)(H - 1) / 2);
        Pm[r][2] = m3;
    }
}

/* ------------------------------------------------------------------------------------------
 * Stage 1: per-surfel preprocess  (upstream preprocessCUDA fwd)
 * Outputs (all per surfel): depth, radius, centre xy, transMat[9] (= T column-major: Tu,Tv,Tw),
 * normal_opacity[4], rgb[3], clamped[3], tile rect -> tiles_touched.
 * ---------------------------------------------------------------------------------------- */
int64_t oracle_preprocess(const oracle_params* prm, const float* means3D, const float* opacities,
                          const float* scales, const float* rotations, const float* transMat_precomp,
                          const float* colors_precomp, const float* shs, const float* viewmatrix,
                          const float* projmatrix, const float* campos,
                          real* depths, int* radii, real* xy, real* transMat, real* normal_opacity, real* rgb,
                          uint8_t* clamped, uint32_t* tiles_touched) {
    const int P = prm->P, W = prm->W, H = prm->H;
    const int gx = (W + TILE - 1) / TILE, gy = (H + TILE - 1) / TILE;
    const real mod = (real)prm->scale_modifier;
    real Pm[4][3];
    world2pix(projmatrix, W, H, Pm);
    const float* vm = viewmatrix;
    int64_t total = 0;

#pragma omp parallel for schedule(static) reduction(+ : total)
    for (int i = 0; i < P; i++) {
        MCA_RESEED(i);
        radii[i] = 0;
        tiles_touched[i] = 0;
        if (g_extent) g_extent[i] = 0;
        depths[i] = 0;
        xy[2 * i] = xy[2 * i + 1] = 0;
        for (int k = 0; k < 4; k++) normal_opacity[4 * i + k] = 0;
        for (int k = 0; k < 3; k++) { clamped[3 * i + k] = 0; if (!colors_precomp) rgb[3 * i + k] = 0; }
        if (!transMat_precomp) for (int k = 0; k < 9; k++) transMat[9 * i + k] = 0;

        const real px = means3D[3 * i], py = means3D[3 * i + 1], pz = means3D[3 * i + 2];
        /* view-space point: viewmatrix is world_view_transform already transposed (cameras.py:56) */
        const real vx = vm[0] * px + vm[4] * py + vm[8] * pz + vm[12];
        const real vy = vm[1] * px + vm[5] * py + vm[9] * pz + vm[13];
        const real vz = vm[2] * px + vm[6] * py + vm[10] * pz + vm[14];
        if (vz <= (real)0.2) continue;                       /* [UPSTREAM-RECALL] near cull */

        real T[9];                                            /* Tu(3) Tv(3) Tw(3) */
        real n[3];
        if (!transMat_precomp) {
            real q[4] = {rotations[4 * i], rotations[4 * i + 1], rotations[4 * i + 2], rotations[4 * i + 3]};
            real R[3][3];
            quat_to_R(q, R);
            const real sx = mod * scales[2 * i], sy = mod * scales[2 * i + 1];
            real L0[3] = {R[0][0] * sx, R[1][0] * sx, R[2][0] * sx};
            real L1[3] = {R[0][1] * sy, R[1][1] * sy, R[2][1] * sy};
            real L2[3] = {R[0][2], R[1][2], R[2][2]};
            for (int c = 0; c < 3; c++) {
                T[3 * c + 0] = L0[0] * Pm[0][c] + L0[1] * Pm[1][c] + L0[2] * Pm[2][c];
                T[3 * c + 1] = L1[0] * Pm[0][c] + L1[1] * Pm[1][c] + L1[2] * Pm[2][c];
                T[3 * c + 2] = px * Pm[0][c] + py * Pm[1][c] + pz * Pm[2][c] + Pm[3][c];
            }
            for (int k = 0; k < 9; k++) transMat[9 * i + k] = T[k];
            n[0] = vm[0] * L2[0] + vm[4] * L2[1] + vm[8] * L2[2];
            n[1] = vm[1] * L2[0] + vm[5] * L2[1] + vm[9] * L2[2];
            n[2] = vm[2] * L2[0] + vm[6] * L2[1] + vm[10] * L2[2];
        } else {
            for (int k = 0; k < 9; k++) T[k] = transMat_precomp[9 * i + k];
            n[0] = 0; n[1] = 0; n[2] = 1;
        }
        /* dual visibility: flip the normal to face the camera [UPSTREAM-RECALL] */
        const real cosv = -(vx * n[0] + vy * n[1] + vz * n[2]);
        if (cosv == 0) continue;
        const real flip = cosv > 0 ? (real)1 : (real)-1;
        n[0] *= flip; n[1] *= flip; n[2] *= flip;

        /* AABB of the 3-sigma ellipse under the homography (paper appendix) */
        const real* Tu = T; const real* Tv = T + 3; const real* Tw = T + 6;
        const real t[3] = {CUTOFF * CUTOFF, CUTOFF * CUTOFF, (real)-1};
        const real dist = t[0] * Tw[0] * Tw[0] + t[1] * Tw[1] * Tw[1] + t[2] * Tw[2] * Tw[2];
        if (dist == 0) continue;
        const real f[3] = {t[0] / dist, t[1] / dist, t[2] / dist};
        const real cx = f[0] * Tu[0] * Tw[0] + f[1] * Tu[1] * Tw[1] + f[2] * Tu[2] * Tw[2];
        const real cy = f[0] * Tv[0] * Tw[0] + f[1] * Tv[1] * Tw[1] + f[2] * Tv[2] * Tw[2];
        const real hx = cx * cx - (f[0] * Tu[0] * Tu[0] + f[1] * Tu[1] * Tu[1] + f[2] * Tu[2] * Tu[2]);
        const real hy = cy * cy - (f[0] * Tv[0] * Tv[0] + f[1] * Tv[1] * Tv[1] + f[2] * Tv[2] * Tv[2]);
        const real ex = R_SQRT(hx > (real)1e-4 ? hx : (real)1e-4);
        const real ey = R_SQRT(hy > (real)1e-4 ? hy : (real)1e-4);
        real rad = ex > ey ? ex : ey;
        if (!(rad > CUTOFF * FILTER_SIZE)) rad = CUTOFF * FILTER_SIZE;
        if (g_extent) g_extent[i] = (double)rad;
        rad = R_CEIL(rad);
        const int irad = (int)rad;
        int x0, y0, x1, y1;
        tile_rect(cx, cy, irad, gx, gy, &x0, &y0, &x1, &y1);
        if ((x1 - x0) * (y1 - y0) == 0) continue;

        if (!colors_precomp) {
            /* SH -> RGB: /root/reference/utils/sh_utils.py:57-112, dir = normalise(mean - campos)
             * gaussian_renderer/__init__.py:88-91; sh layout [P, M, 3] (gaussian_model.py:108-112) */
            const float* sh = shs + (size_t)i * prm->M * 3;
            real dx = px - campos[0], dy = py - campos[1], dz = pz - campos[2];
            const real il = 1 / R_SQRT(dx * dx + dy * dy + dz * dz);
            dx *= il; dy *= il; dz *= il;
            for (int c = 0; c < 3; c++) {
                real r = SH_C0 * sh[c];
                if (prm->D > 0) {
                    r = r - SH_C1 * dy * sh[3 + c] + SH_C1 * dz * sh[6 + c] - SH_C1 * dx * sh[9 + c];
                    if (prm->D > 1) {
                        real xx = dx * dx, yy = dy * dy, zz = dz * dz, xy_ = dx * dy, yz = dy * dz, xz = dx * dz;
                        r = r + SH_C2[0] * xy_ * sh[12 + c] + SH_C2[1] * yz * sh[15 + c] +
                            SH_C2[2] * (2 * zz - xx - yy) * sh[18 + c] + SH_C2[3] * xz * sh[21 + c] +
                            SH_C2[4] * (xx - yy) * sh[24 + c];
                        if (prm->D > 2) {
                            r = r + SH_C3[0] * dy * (3 * xx - yy) * sh[27 + c] + SH_C3[1] * xy_ * dz * sh[30 + c] +
                                SH_C3[2] * dy * (4 * zz - xx - yy) * sh[33 + c] +
                                SH_C3[3] * dz * (2 * zz - 3 * xx - 3 * yy) * sh[36 + c] +
                                SH_C3[4] * dx * (4 * zz - xx - yy) * sh[39 + c] +
                                SH_C3[5] * dz * (xx - yy) * sh[42 + c] + SH_C3[6] * dx * (xx - 3 * yy) * sh[45 + c];
                        }
                    }
                }
                r += (real)0.5;
                clamped[3 * i + c] = r < 0;
                rgb[3 * i + c] = r < 0 ? 0 : r;
            }
        }
        depths[i] = vz;
        radii[i] = irad;
        xy[2 * i] = cx; xy[2 * i + 1] = cy;
        normal_opacity[4 * i + 0] = n[0]; normal_opacity[4 * i + 1] = n[1]; normal_opacity[4 * i + 2] = n[2];
        normal_opacity[4 * i + 3] = opacities[i];
        tiles_touched[i] = (uint32_t)((x1 - x0) * (y1 - y0));
        total += (x1 - x0) * (y1 - y0);
    }
    return total;
}

/* ------------------------------------------------------------------------------------------
 * Stage 2: binning.  Emit (tile, depth-bits) keys in surfel order, stable sort, tile ranges.
 * [UPSTREAM-RECALL] key = tile_id << 32 | float32 bits of view depth; stable LSD radix sort.
 * `depth_f32` lets the caller inject the float32 depths of the device run so both sides sort
 * on identical keys; pass NULL to use (float)depths[i].
 * ---------------------------------------------------------------------------------------- */
typedef struct { uint32_t bits; uint32_t val; } kv_t;
static int kv_cmp(const void* a, const void* b) {
    const kv_t* x = (const kv_t*)a; const kv_t* y = (const kv_t*)b;
    if (x->bits != y->bits) return x->bits < y->bits ? -1 : 1;
    return x->val < y->val ? -1 : (x->val > y->val);
}

/* The global stable sort on (tile << 32 | depth bits) with emission order (surfel-major) as the tie-break is the same
 * order as: bucket the instances by tile, then order every tile's bucket by (depth bits, surfel index) — a surfel appears
 * at most once per tile.  Done that way the per-tile sorts run in parallel (the C4 / C5 parity runs hold 1e7 - 1e8 instances). */
int oracle_bin(const oracle_params* prm, const real* depths, const float* depth_f32, const int* radii, const real* xy,
               int64_t R, uint32_t* point_list, uint64_t* keys_sorted, uint32_t* ranges /* [tiles][2] */) {
    const int P = prm->P, W = prm->W, H = prm->H;
    const int gx = (W + TILE - 1) / TILE, gy = (H + TILE - 1) / TILE;
    const int ntiles = gx * gy;
    kv_t* kv = (kv_t*)malloc(sizeof(kv_t) * (size_t)(R > 0 ? R : 1));
    int64_t* start = (int64_t*)calloc((size_t)ntiles + 1, sizeof(int64_t));
    int64_t* fill = (int64_t*)malloc(sizeof(int64_t) * ((size_t)ntiles + 1));
    if (!kv || !start || !fill) { free(kv); free(start); free(fill); return -1; }
    int64_t total = 0;
    for (int i = 0; i < P; i++) {
        if (radii[i] <= 0) continue;
        int x0, y0, x1, y1;
        tile_rect(xy[2 * i], xy[2 * i + 1], radii[i], gx, gy, &x0, &y0, &x1, &y1);
        for (int y = y0; y < y1; y++)
            for (int x = x0; x < x1; x++) { start[y * gx + x + 1]++; total++; }
    }
    if (total != R) { free(kv); free(start); free(fill); return -3; }
    for (int t = 0; t < ntiles; t++) start[t + 1] += start[t];
    memcpy(fill, start, sizeof(int64_t) * ((size_t)ntiles + 1));
    for (int i = 0; i < P; i++) {
        if (radii[i] <= 0) continue;
        int x0, y0, x1, y1;
        tile_rect(xy[2 * i], xy[2 * i + 1], radii[i], gx, gy, &x0, &y0, &x1, &y1);
        float d = depth_f32 ? depth_f32[i] : (float)depths[i];
        uint32_t bits; memcpy(&bits, &d, 4);
        for (int y = y0; y < y1; y++)
            for (int x = x0; x < x1; x++) {
                const int64_t k = fill[y * gx + x]++;
                kv[k].bits = bits; kv[k].val = (uint32_t)i;
            }
    }
    memset(ranges, 0, sizeof(uint32_t) * 2 * (size_t)ntiles);
#pragma omp parallel for schedule(dynamic, 4)
    for (int t = 0; t < ntiles; t++) {
        const int64_t b = start[t], e = start[t + 1];
        if (e == b) continue;
        qsort(kv + b, (size_t)(e - b), sizeof(kv_t), kv_cmp);
        ranges[2 * t] = (uint32_t)b; ranges[2 * t + 1] = (uint32_t)e;
        for (int64_t k = b; k < e; k++) {
            point_list[k] = kv[k].val;
            if (keys_sorted) keys_sorted[k] = ((uint64_t)(uint32_t)t << 32) | kv[k].bits;
        }
    }
    free(kv); free(start); free(fill);
    return 0;
}

/* ray–splat intersection + alpha for one (pixel, surfel) pair.  Returns 0 if the pair is
 * skipped.  Paper eqs. 8-11; thresholds [UPSTREAM-RECALL]. */
typedef struct {
    real sx, sy, pz, kx, ky, kz, lx, ly, lz, dx, dy, rho3d, rho2d, depth, G, alpha;
} hit_t;

static inline int intersect(const real* T, const real* cxy, real opa, real pxf, real pyf, hit_t* h) {
    const real* Tu = T; const real* Tv = T + 3; const real* Tw = T + 6;
    h->kx = pxf * Tw[0] - Tu[0]; h->ky = pxf * Tw[1] - Tu[1]; h->kz = pxf * Tw[2] - Tu[2];
    h->lx = pyf * Tw[0] - Tv[0]; h->ly = pyf * Tw[1] - Tv[1]; h->lz = pyf * Tw[2] - Tv[2];
    const real p0 = h->ky * h->lz - h->kz * h->ly;
    const real p1 = h->kz * h->lx - h->kx * h->lz;
    const real p2 = h->kx * h->ly - h->ky * h->lx;
    if (p2 == 0) return 0;
    h->pz = p2;
    h->sx = p0 / p2; h->sy = p1 / p2;
    h->rho3d = h->sx * h->sx + h->sy * h->sy;
    h->dx = cxy[0] - pxf; h->dy = cxy[1] - pyf;
    h->rho2d = FILTER_INV_SQUARE * (h->dx * h->dx + h->dy * h->dy);
    const real rho = h->rho3d < h->rho2d ? h->rho3d : h->rho2d;
    h->depth = (h->rho3d <= h->rho2d) ? (h->sx * Tw[0] + h->sy * Tw[1]) + Tw[2] : Tw[2];
    if (h->depth < NEAR_N) return 0;
    const real power = (real)-0.5 * rho;
    if (power > 0) return 0;
    h->G = R_EXP(power);
    real a = opa * h->G;
    h->alpha = a < ALPHA_MAX ? a : ALPHA_MAX;
    if (h->alpha < ALPHA_MIN) return 0;
    return 1;
}

/* ------------------------------------------------------------------------------------------
 * Stage 3: per-pixel front-to-back blend (upstream renderCUDA fwd).
 * out_color[3,H,W]; out_others[7,H,W] with the channel contract pinned by
 * /root/reference/gaussian_renderer/__init__.py:118-135: 0 = sum w*depth, 1 = 1-T, 2..4 = sum
 * w*normal(view), 5 = median depth, 6 = distortion.  State for backward: final_T[3,H,W] =
 * (T, M1, M2), n_contrib[2,H,W] = (last contributor, median contributor), 1-based list positions.
 * ---------------------------------------------------------------------------------------- */
void oracle_blend_forward(const oracle_params* prm, const uint32_t* ranges, const uint32_t* point_list,
                          const real* xy, const real* transMat, const float* transMat_precomp,
                          const real* normal_opacity, const real* rgb, const float* colors_precomp,
                          const float* bg, real* out_color, real* out_others, real* final_T, uint32_t* n_contrib) {
    const int W = prm->W, H = prm->H;
    const int gx = (W + TILE - 1) / TILE, gy = (H + TILE - 1) / TILE;
    const size_t HW = (size_t)H * W;
#pragma omp parallel for schedule(dynamic, 1)
    for (int tile = 0; tile < gx * gy; tile++) {
        const uint32_t r0 = ranges[2 * tile], r1 = ranges[2 * tile + 1];
        const int bx = (tile % gx) * TILE, by = (tile / gx) * TILE;
        MCA_RESEED(0x100000000ull + (uint64_t)tile);
        for (int ty = 0; ty < TILE; ty++)
            for (int tx = 0; tx < TILE; tx++) {
                const int pxi = bx + tx, pyi = by + ty;
                if (pxi >= W || pyi >= H) continue;
                const size_t pix = (size_t)pyi * W + pxi;
                const real pxf = (real)pxi, pyf = (real)pyi;
                uint64_t sigp = 0;
                uint32_t med_id = 0xffffffffu;
                real T = 1, C[3] = {0, 0, 0}, N[3] = {0, 0, 0}, D = 0, M1 = 0, M2 = 0, dist = 0, med = 0;
                uint32_t contributor = 0, last = 0, medc = 0;
                for (uint32_t k = r0; k < r1; k++) {
                    contributor++;
                    const uint32_t id = point_list[k];
                    real Tm[9];
                    if (transMat_precomp) for (int q = 0; q < 9; q++) Tm[q] = transMat_precomp[9 * (size_t)id + q];
                    else for (int q = 0; q < 9; q++) Tm[q] = transMat[9 * (size_t)id + q];
                    hit_t h;
                    if (!intersect(Tm, xy + 2 * (size_t)id, normal_opacity[4 * (size_t)id + 3], pxf, pyf, &h)) continue;
                    const real testT = T * (1 - h.alpha);
                    if (testT < T_EPS) break;     /* terminating surfel is NOT composited */
                    const real w = h.alpha * T;
                    const real A = 1 - T;
                    const real m = FAR_N / (FAR_N - NEAR_N) * (1 - NEAR_N / h.depth);
                    dist += (m * m * A + M2 - 2 * m * M1) * w;       /* paper eq. 17, O(N) form */
                    D += h.depth * w;
                    M1 += m * w;
                    M2 += m * m * w;
                    if (T > (real)0.5) { med = h.depth; medc = contributor; med_id = id; }
                    if (g_sig_pix) {
                        const uint64_t br = (h.rho3d <= h.rho2d) ? 1u : 0u;
                        sigp += sig_mix(((uint64_t)id << 1) | br);
                        const uint64_t ss = sig_mix(((uint64_t)pix << 1) | br);
#pragma omp atomic
                        g_sig_surf[id] += ss;
                    }
                    for (int c = 0; c < 3; c++) N[c] += normal_opacity[4 * (size_t)id + c] * w;
                    for (int c = 0; c < 3; c++) {
                        const real col = colors_precomp ? (real)colors_precomp[3 * (size_t)id + c] : rgb[3 * (size_t)id + c];
                        C[c] += col * w;
                    }
                    T = testT;
                    last = contributor;
                }
                if (g_sig_pix) {
                    g_sig_pix[pix] = sigp + (med_id != 0xffffffffu ? sig_mix(0x4D454400000000ull + med_id) : 0ull);
                    if (med_id != 0xffffffffu) {
                        const uint64_t ss = sig_mix(0x4D454400000000ull + (uint64_t)pix);
#pragma omp atomic
                        g_sig_surf[med_id] += ss;
                    }
                }
                final_T[pix] = T; final_T[HW + pix] = M1; final_T[2 * HW + pix] = M2;
                n_contrib[pix] = last; n_contrib[HW + pix] = medc;
                for (int c = 0; c < 3; c++) out_color[c * HW + pix] = C[c] + T * bg[c];
                out_others[0 * HW + pix] = D;
                out_others[1 * HW + pix] = 1 - T;
                for (int c = 0; c < 3; c++) out_others[(2 + c) * HW + pix] = N[c];
                out_others[5 * HW + pix] = med;
                out_others[6 * HW + pix] = dist;
            }
    }
}

/* ------------------------------------------------------------------------------------------
 * Stage 4: per-pixel back-to-front backward (upstream renderCUDA bwd).
 * Accumulates dL/d{transMat[9], mean2D[2] (low-pass branch), normal[3], opacity, rgb[3]}.
 * Tiles are processed serially per surfel accumulator (OpenMP atomics) so fp64 sums are
 * order-insensitive to ~1e-16.
 * ---------------------------------------------------------------------------------------- */
void oracle_blend_backward(const oracle_params* prm, const uint32_t* ranges, const uint32_t* point_list,
                           const real* xy, const real* transMat, const float* transMat_precomp,
                           const real* normal_opacity, const real* rgb, const float* colors_precomp,
                           const float* bg, const real* final_T, const uint32_t* n_contrib,
                           const real* dL_dpix /*[3,H,W]*/, const real* dL_dothers /*[7,H,W]*/,
                           real* dL_dtransMat /*[P,9]*/, real* dL_dmean2D /*[P,3]*/, real* dL_dnormal /*[P,3]*/,
                           real* dL_dopacity /*[P]*/, real* dL_dcolors /*[P,3]*/) {
    const int W = prm->W, H = prm->H;
    const int gx = (W + TILE - 1) / TILE, gy = (H + TILE - 1) / TILE;
    const size_t HW = (size_t)H * W;
#pragma omp parallel for schedule(dynamic, 1)
    for (int tile = 0; tile < gx * gy; tile++) {
        const uint32_t r0 = ranges[2 * tile], r1 = ranges[2 * tile + 1];
        const int bx = (tile % gx) * TILE, by = (tile / gx) * TILE;
        MCA_RESEED(0x200000000ull + (uint64_t)tile);
        for (int ty = 0; ty < TILE; ty++)
            for (int tx = 0; tx < TILE; tx++) {
                const int pxi = bx + tx, pyi = by + ty;
                if (pxi >= W || pyi >= H) continue;
                const size_t pix = (size_t)pyi * W + pxi;
                const real pxf = (real)pxi, pyf = (real)pyi;
                const real T_final = final_T[pix];
                const real final_M1 = final_T[HW + pix], final_M2 = final_T[2 * HW + pix];
                const real final_A = 1 - T_final;
                const uint32_t last = n_contrib[pix], medc = n_contrib[HW + pix];
                real gC[3] = {dL_dpix[pix], dL_dpix[HW + pix], dL_dpix[2 * HW + pix]};
                const real g_depth = dL_dothers[0 * HW + pix], g_alpha = dL_dothers[1 * HW + pix];
                const real gN[3] = {dL_dothers[2 * HW + pix], dL_dothers[3 * HW + pix], dL_dothers[4 * HW + pix]};
                const real g_med = dL_dothers[5 * HW + pix], g_dist = dL_dothers[6 * HW + pix];
                const real bg_dot = bg[0] * gC[0] + bg[1] * gC[1] + bg[2] * gC[2];

                real T = T_final;
                real last_alpha = 0, last_color[3] = {0, 0, 0}, accum_rec[3] = {0, 0, 0};
                real last_depth = 0, accum_depth = 0, accum_alpha = 0;
                real last_normal[3] = {0, 0, 0}, accum_normal[3] = {0, 0, 0};
                real last_dL_dT = 0;
                /* walk list positions last..1 (1-based), i.e. k = r0+last-1 down to r0 */
                for (uint32_t c = last; c >= 1; c--) {
                    const uint32_t k = r0 + c - 1;
                    (void)r1;
                    const uint32_t id = point_list[k];
                    real Tm[9];
                    if (transMat_precomp) for (int q = 0; q < 9; q++) Tm[q] = transMat_precomp[9 * (size_t)id + q];
                    else for (int q = 0; q < 9; q++) Tm[q] = transMat[9 * (size_t)id + q];
                    const real* Tw = Tm + 6;
                    const real opa = normal_opacity[4 * (size_t)id + 3];
                    hit_t h;
                    if (!intersect(Tm, xy + 2 * (size_t)id, opa, pxf, pyf, &h)) continue;
                    const real alpha = h.alpha, G = h.G;
                    T = T / (1 - alpha);
                    const real w = alpha * T;
                    real dL_dalpha = 0;
                    real gcol[3];
                    for (int ch = 0; ch < 3; ch++) {
                        const real col = colors_precomp ? (real)colors_precomp[3 * (size_t)id + ch] : rgb[3 * (size_t)id + ch];
                        accum_rec[ch] = last_alpha * last_color[ch] + (1 - last_alpha) * accum_rec[ch];
                        last_color[ch] = col;
                        dL_dalpha += (col - accum_rec[ch]) * gC[ch];
                        gcol[ch] = w * gC[ch];
                    }
                    real dL_dz = 0;
                    const real m = FAR_N / (FAR_N - NEAR_N) * (1 - NEAR_N / h.depth);
                    const real dm_dd = (FAR_N * NEAR_N) / ((FAR_N - NEAR_N) * h.depth * h.depth);
                    if (c == medc) dL_dz += g_med;                 /* median depth: selected surfel only */
                    const real dL_dweight = (final_M2 + m * m * final_A - 2 * m * final_M1) * g_dist;
                    dL_dalpha += dL_dweight - last_dL_dT;
                    last_dL_dT = dL_dweight * alpha + (1 - alpha) * last_dL_dT;
                    const real dL_dm = 2 * w * (m * final_A - final_M1) * g_dist;
                    dL_dz += dL_dm * dm_dd;
                    accum_depth = last_alpha * last_depth + (1 - last_alpha) * accum_depth;
                    last_depth = h.depth;
                    dL_dalpha += (h.depth - accum_depth) * g_depth;
                    accum_alpha = last_alpha + (1 - last_alpha) * accum_alpha;
                    dL_dalpha += (1 - accum_alpha) * g_alpha;
                    real gnor[3];
                    for (int ch = 0; ch < 3; ch++) {
                        const real nn = normal_opacity[4 * (size_t)id + ch];
                        accum_normal[ch] = last_alpha * last_normal[ch] + (1 - last_alpha) * accum_normal[ch];
                        last_normal[ch] = nn;
                        dL_dalpha += (nn - accum_normal[ch]) * gN[ch];
                        gnor[ch] = w * gN[ch];
                    }
                    dL_dalpha *= T;
                    last_alpha = alpha;
                    dL_dalpha += (-T_final / (1 - alpha)) * bg_dot;
                    /* the 0.99 clamp is treated as pass-through [UPSTREAM-RECALL] */
                    const real dL_dG = opa * dL_dalpha;
                    dL_dz += w * g_depth;

                    real gT[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, gxy[2] = {0, 0};
                    if (h.rho3d <= h.rho2d) {
                        const real dsx = dL_dG * -G * h.sx + dL_dz * Tw[0];
                        const real dsy = dL_dG * -G * h.sy + dL_dz * Tw[1];
                        const real ax = dsx / h.pz, ay = dsy / h.pz;
                        const real dp[3] = {ax, ay, -(ax * h.sx + ay * h.sy)};
                        /* p = k x l  ->  dk = l x dp, dl = dp x k */
                        const real dk[3] = {h.ly * dp[2] - h.lz * dp[1], h.lz * dp[0] - h.lx * dp[2], h.lx * dp[1] - h.ly * dp[0]};
                        const real dl[3] = {dp[1] * h.kz - dp[2] * h.ky, dp[2] * h.kx - dp[0] * h.kz, dp[0] * h.ky - dp[1] * h.kx};
                        gT[0] = -dk[0]; gT[1] = -dk[1]; gT[2] = -dk[2];
                        gT[3] = -dl[0]; gT[4] = -dl[1]; gT[5] = -dl[2];
                        gT[6] = pxf * dk[0] + pyf * dl[0] + dL_dz * h.sx;
                        gT[7] = pxf * dk[1] + pyf * dl[1] + dL_dz * h.sy;
                        gT[8] = pxf * dk[2] + pyf * dl[2] + dL_dz;
                    } else {
                        gxy[0] = dL_dG * (-G * FILTER_INV_SQUARE * h.dx);
                        gxy[1] = dL_dG * (-G * FILTER_INV_SQUARE * h.dy);
                        gT[8] = dL_dz;
                    }
                    const real gop = G * dL_dalpha;
                    for (int q = 0; q < 9; q++) if (gT[q] != 0) ATOMIC_ADD(dL_dtransMat[9 * (size_t)id + q], gT[q]);
                    for (int q = 0; q < 2; q++) if (gxy[q] != 0) ATOMIC_ADD(dL_dmean2D[3 * (size_t)id + q], gxy[q]);
                    for (int q = 0; q < 3; q++) {
                        ATOMIC_ADD(dL_dnormal[3 * (size_t)id + q], gnor[q]);
                        ATOMIC_ADD(dL_dcolors[3 * (size_t)id + q], gcol[q]);
                    }
                    ATOMIC_ADD(dL_dopacity[id], gop);
                }
            }
    }
}

/* ------------------------------------------------------------------------------------------
 * Stage 5: per-surfel preprocess backward (upstream preprocessCUDA bwd).
 * In: dL_dtransMat, dL_dmean2D (low-pass-branch gradient), dL_dnormal, dL_dcolors.
 * Out: dL_dmeans3D[P,3], dL_dscales[P,2], dL_drots[P,4], dL_dsh[P,M,3]; dL_dmean2D is
 * OVERWRITTEN with the densification statistic (gaussian_model.py:405-407 consumes its norm);
 * with transMat_precomp the low-pass term is folded into dL_dtransMat instead.
 * ---------------------------------------------------------------------------------------- */
void oracle_preprocess_backward(const oracle_params* prm, const float* means3D, const int* radii,
                                const float* shs, const uint8_t* clamped, const float* scales,
                                const float* rotations, const float* transMat_precomp, const real* transMat,
                                const float* viewmatrix, const float* projmatrix, const float* campos,
                                real* dL_dtransMat, const real* dL_dnormal, const real* dL_dcolors,
                                real* dL_dsh, real* dL_dmean2D, real* dL_dmeans3D, real* dL_dscales, real* dL_drots) {
    const int P = prm->P, W = prm->W, H = prm->H;
    const real mod = (real)prm->scale_modifier;
    real Pm[4][3];
    world2pix(projmatrix, W, H, Pm);
    const float* vm = viewmatrix;
#pragma omp parallel for schedule(static)
    for (int i = 0; i < P; i++) {
        MCA_RESEED(0x300000000ull + (uint64_t)i);
        if (!(radii[i] > 0)) continue;
        real T[9];
        if (transMat_precomp) for (int k = 0; k < 9; k++) T[k] = transMat_precomp[9 * i + k];
        else for (int k = 0; k < 9; k++) T[k] = transMat[9 * i + k];
        real* g = dL_dtransMat + 9 * (size_t)i;           /* g[3c+r] = dL/dT_math[r][c] */
        real gT[9];
        for (int k = 0; k < 9; k++) gT[k] = g[k];
        const real gx2 = dL_dmean2D[3 * i], gy2 = dL_dmean2D[3 * i + 1];
        if (gx2 != 0 || gy2 != 0) {
            /* gradient of the AABB centre (the low-pass filter's centre) w.r.t. T */
            const real* Tu = T; const real* Tv = T + 3; const real* Tw = T + 6;
            const real t[3] = {CUTOFF * CUTOFF, CUTOFF * CUTOFF, (real)-1};
            const real d = t[0] * Tw[0] * Tw[0] + t[1] * Tw[1] * Tw[1] + t[2] * Tw[2] * Tw[2];
            real f[3], dL_df[3];
            for (int k = 0; k < 3; k++) f[k] = t[k] / d;
            real dd = 0;
            for (int k = 0; k < 3; k++) {
                dL_df[k] = gx2 * Tu[k] * Tw[k] + gy2 * Tv[k] * Tw[k];
                dd += dL_df[k] * f[k];
            }
            dd *= -1 / d;
            for (int k = 0; k < 3; k++) {
                gT[k] += gx2 * f[k] * Tw[k];
                gT[3 + k] += gy2 * f[k] * Tw[k];
                gT[6 + k] += gx2 * f[k] * Tu[k] + gy2 * f[k] * Tv[k] + dd * t[k] * Tw[k] * 2;
            }
            if (transMat_precomp) for (int k = 0; k < 9; k++) g[k] = gT[k];
        }
        if (!transMat_precomp) {
            const real px = means3D[3 * i], py = means3D[3 * i + 1], pz = means3D[3 * i + 2];
            real q[4] = {rotations[4 * i], rotations[4 * i + 1], rotations[4 * i + 2], rotations[4 * i + 3]};
            real R[3][3];
            quat_to_R(q, R);
            const real sx = mod * scales[2 * i], sy = mod * scales[2 * i + 1];
            /* dL/dA[r][j] = sum_c gT[3c+r] * Pm[j][c]  (T_math = A * Pm) */
            real dA[3][4];
            for (int r = 0; r < 3; r++)
                for (int j = 0; j < 4; j++) dA[r][j] = gT[r] * Pm[j][0] + gT[3 + r] * Pm[j][1] + gT[6 + r] * Pm[j][2];
            /* normal: n_view = flip * V3x3 * R[:,2] */
            const real L2[3] = {R[0][2], R[1][2], R[2][2]};
            real n[3] = {vm[0] * L2[0] + vm[4] * L2[1] + vm[8] * L2[2], vm[1] * L2[0] + vm[5] * L2[1] + vm[9] * L2[2],
                         vm[2] * L2[0] + vm[6] * L2[1] + vm[10] * L2[2]};
            const real vx = vm[0] * px + vm[4] * py + vm[8] * pz + vm[12];
            const real vy = vm[1] * px + vm[5] * py + vm[9] * pz + vm[13];
            const real vz = vm[2] * px + vm[6] * py + vm[10] * pz + vm[14];
            const real flip = (-(vx * n[0] + vy * n[1] + vz * n[2])) > 0 ? (real)1 : (real)-1;
            const real* gn = dL_dnormal + 3 * (size_t)i;
            real dtn[3] = {flip * (vm[0] * gn[0] + vm[1] * gn[1] + vm[2] * gn[2]),
                           flip * (vm[4] * gn[0] + vm[5] * gn[1] + vm[6] * gn[2]),
                           flip * (vm[8] * gn[0] + vm[9] * gn[1] + vm[10] * gn[2])};
            /* dL/dR[r][c]: columns 0,1 scaled, column 2 from the normal */
            real dR[3][3];
            for (int r = 0; r < 3; r++) { dR[r][0] = dA[0][r] * sx; dR[r][1] = dA[1][r] * sy; dR[r][2] = dtn[r]; }
            dL_dscales[2 * i + 0] = mod * (dA[0][0] * R[0][0] + dA[0][1] * R[1][0] + dA[0][2] * R[2][0]);
            dL_dscales[2 * i + 1] = mod * (dA[1][0] * R[0][1] + dA[1][1] * R[1][1] + dA[1][2] * R[2][1]);
            dL_dmeans3D[3 * i + 0] += dA[2][0];
            dL_dmeans3D[3 * i + 1] += dA[2][1];
            dL_dmeans3D[3 * i + 2] += dA[2][2];
            /* vjp through q -> R for the *normalised* quaternion (no projection through the
             * normalisation: the caller's F.normalize backward applies it, gaussian_model.py:41) */
            const real s = 1 / R_SQRT(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
            const real w = q[0] * s, x = q[1] * s, y = q[2] * s, z = q[3] * s;
            dL_drots[4 * i + 0] = 2 * (x * (dR[2][1] - dR[1][2]) + y * (dR[0][2] - dR[2][0]) + z * (dR[1][0] - dR[0][1]));
            dL_drots[4 * i + 1] = 2 * (-2 * x * (dR[1][1] + dR[2][2]) + y * (dR[0][1] + dR[1][0]) + z * (dR[0][2] + dR[2][0]) + w * (dR[2][1] - dR[1][2]));
            dL_drots[4 * i + 2] = 2 * (x * (dR[0][1] + dR[1][0]) - 2 * y * (dR[0][0] + dR[2][2]) + z * (dR[1][2] + dR[2][1]) + w * (dR[0][2] - dR[2][0]));
            dL_drots[4 * i + 3] = 2 * (x * (dR[0][2] + dR[2][0]) + y * (dR[1][2] + dR[2][1]) - 2 * z * (dR[0][0] + dR[1][1]) + w * (dR[1][0] - dR[0][1]));
        }
        if (shs) {
            /* SH backward incl. the view-direction path into the mean (3DGS lineage) */
            const float* sh = shs + (size_t)i * prm->M * 3;
            real* gsh = dL_dsh + (size_t)i * prm->M * 3;
            const real px = means3D[3 * i], py = means3D[3 * i + 1], pz = means3D[3 * i + 2];
            real dox = px - campos[0], doy = py - campos[1], doz = pz - campos[2];
            const real il = 1 / R_SQRT(dox * dox + doy * doy + doz * doz);
            const real x = dox * il, y = doy * il, z = doz * il;
            real gRGB[3];
            for (int c = 0; c < 3; c++) gRGB[c] = clamped[3 * i + c] ? 0 : dL_dcolors[3 * (size_t)i + c];
            real dRGBdx[3] = {0, 0, 0}, dRGBdy[3] = {0, 0, 0}, dRGBdz[3] = {0, 0, 0};
            for (int c = 0; c < 3; c++) gsh[c] = SH_C0 * gRGB[c];
            if (prm->D > 0) {
                for (int c = 0; c < 3; c++) {
                    gsh[3 + c] = -SH_C1 * y * gRGB[c];
                    gsh[6 + c] = SH_C1 * z * gRGB[c];
                    gsh[9 + c] = -SH_C1 * x * gRGB[c];
                    dRGBdx[c] = -SH_C1 * sh[9 + c];
                    dRGBdy[c] = -SH_C1 * sh[3 + c];
                    dRGBdz[c] = SH_C1 * sh[6 + c];
                }
                if (prm->D > 1) {
                    const real xx = x * x, yy = y * y, zz = z * z, xy_ = x * y, yz = y * z, xz = x * z;
                    for (int c = 0; c < 3; c++) {
                        gsh[12 + c] = SH_C2[0] * xy_ * gRGB[c];
                        gsh[15 + c] = SH_C2[1] * yz * gRGB[c];
                        gsh[18 + c] = SH_C2[2] * (2 * zz - xx - yy) * gRGB[c];
                        gsh[21 + c] = SH_C2[3] * xz * gRGB[c];
                        gsh[24 + c] = SH_C2[4] * (xx - yy) * gRGB[c];
                        dRGBdx[c] += SH_C2[0] * y * sh[12 + c] + SH_C2[2] * 2 * -x * sh[18 + c] + SH_C2[3] * z * sh[21 + c] + SH_C2[4] * 2 * x * sh[24 + c];
                        dRGBdy[c] += SH_C2[0] * x * sh[12 + c] + SH_C2[1] * z * sh[15 + c] + SH_C2[2] * 2 * -y * sh[18 + c] + SH_C2[4] * 2 * -y * sh[24 + c];
                        dRGBdz[c] += SH_C2[1] * y * sh[15 + c] + SH_C2[2] * 2 * 2 * z * sh[18 + c] + SH_C2[3] * x * sh[21 + c];
                    }
                    if (prm->D > 2) {
                        for (int c = 0; c < 3; c++) {
                            gsh[27 + c] = SH_C3[0] * y * (3 * xx - yy) * gRGB[c];
                            gsh[30 + c] = SH_C3[1] * xy_ * z * gRGB[c];
                            gsh[33 + c] = SH_C3[2] * y * (4 * zz - xx - yy) * gRGB[c];
                            gsh[36 + c] = SH_C3[3] * z * (2 * zz - 3 * xx - 3 * yy) * gRGB[c];
                            gsh[39 + c] = SH_C3[4] * x * (4 * zz - xx - yy) * gRGB[c];
                            gsh[42 + c] = SH_C3[5] * z * (xx - yy) * gRGB[c];
                            gsh[45 + c] = SH_C3[6] * x * (xx - 3 * yy) * gRGB[c];
                            dRGBdx[c] += SH_C3[0] * sh[27 + c] * 3 * 2 * xy_ + SH_C3[1] * sh[30 + c] * yz +
                                         SH_C3[2] * sh[33 + c] * -2 * xy_ + SH_C3[3] * sh[36 + c] * -3 * 2 * xz +
                                         SH_C3[4] * sh[39 + c] * (-3 * xx + 4 * zz - yy) + SH_C3[5] * sh[42 + c] * 2 * xz +
                                         SH_C3[6] * sh[45 + c] * 3 * (xx - yy);
                            dRGBdy[c] += SH_C3[0] * sh[27 + c] * 3 * (xx - yy) + SH_C3[1] * sh[30 + c] * xz +
                                         SH_C3[2] * sh[33 + c] * (-3 * yy + 4 * zz - xx) + SH_C3[3] * sh[36 + c] * -3 * 2 * yz +
                                         SH_C3[4] * sh[39 + c] * -2 * xy_ + SH_C3[5] * sh[42 + c] * -2 * yz +
                                         SH_C3[6] * sh[45 + c] * -3 * 2 * xy_;
                            dRGBdz[c] += SH_C3[1] * sh[30 + c] * xy_ + SH_C3[2] * sh[33 + c] * 4 * 2 * yz +
                                         SH_C3[3] * sh[36 + c] * 3 * (2 * zz - xx - yy) + SH_C3[4] * sh[39 + c] * 4 * 2 * xz +
                                         SH_C3[5] * sh[42 + c] * (xx - yy);
                        }
                    }
                }
            }
            const real gdx = dRGBdx[0] * gRGB[0] + dRGBdx[1] * gRGB[1] + dRGBdx[2] * gRGB[2];
            const real gdy = dRGBdy[0] * gRGB[0] + dRGBdy[1] * gRGB[1] + dRGBdy[2] * gRGB[2];
            const real gdz = dRGBdz[0] * gRGB[0] + dRGBdz[1] * gRGB[1] + dRGBdz[2] * gRGB[2];
            /* d normalise(v)/dv applied to (gdx,gdy,gdz) */
            const real sum2 = dox * dox + doy * doy + doz * doz;
            const real il3 = 1 / R_SQRT(sum2 * sum2 * sum2);
            dL_dmeans3D[3 * i + 0] += ((sum2 - dox * dox) * gdx - doy * dox * gdy - doz * dox * gdz) * il3;
            dL_dmeans3D[3 * i + 1] += (-dox * doy * gdx + (sum2 - doy * doy) * gdy - doz * doy * gdz) * il3;
            dL_dmeans3D[3 * i + 2] += (-dox * doz * gdx - doy * doz * gdy + (sum2 - doz * doz) * gdz) * il3;
        }
        /* densification statistic replaces the screen-space mean gradient [UPSTREAM-RECALL]:
         * d/d(x_ndc) of the splat translation = dL/dTu.z * depth * W/2 (low-pass gradient excluded,
         * /root/reference/README.md:118). Uses the blend-stage dL_dT (before the centre fold). */
        const real depth = T[8];
        const real g2 = transMat_precomp ? g[2] : dL_dtransMat[9 * (size_t)i + 2];
        const real g5 = transMat_precomp ? g[5] : dL_dtransMat[9 * (size_t)i + 5];
        dL_dmean2D[3 * i + 0] = g2 * depth * (real)0.5 * (real)W;
        dL_dmean2D[3 * i + 1] = g5 * depth * (real)0.5 * (real)H;
    }
}

/* visibility test used by GaussianRasterizer.markVisible [UPSTREAM-RECALL] */
void oracle_mark_visible(int P, const float* means3D, const float* viewmatrix, uint8_t* present) {
    const float* vm = viewmatrix;
    for (int i = 0; i < P; i++) {
        const real vz = (real)vm[2] * means3D[3 * i] + (real)vm[6] * means3D[3 * i + 1] + (real)vm[10] * means3D[3 * i + 2] + vm[14];
        present[i] = vz > (real)0.2;
    }
}

/* ------------------------------------------------------------------------------------------
 * simple-knn: mean squared distance to the 3 nearest neighbours (distCUDA2,
 * /root/reference/scene/gaussian_model.py:134).  Brute force O(P^2), exact.
 * ---------------------------------------------------------------------------------------- */
void oracle_knn_dist2(int P, const float* pts, real* out) {
#pragma omp parallel for schedule(static)
    for (int i = 0; i < P; i++) {
        real best[3] = {(real)3.4028234663852886e38, (real)3.4028234663852886e38, (real)3.4028234663852886e38};
        for (int j = 0; j < P; j++) {
            if (j == i) continue;
            real dx = (real)pts[3 * i] - pts[3 * j], dy = (real)pts[3 * i + 1] - pts[3 * j + 1], dz = (real)pts[3 * i + 2] - pts[3 * j + 2];
            real d = dx * dx + dy * dy + dz * dz;
            for (int k = 0; k < 3; k++)
                if (d < best[k]) { real t = best[k]; best[k] = d; d = t; }
        }
        out[i] = (best[0] + best[1] + best[2]) / 3;
    }
}
#ifdef __cplusplus
}
#endif
