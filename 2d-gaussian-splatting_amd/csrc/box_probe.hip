// box_probe.hip — what THIS GPU box sustains, measured in the bench process right before the timed window (gfx950).
//
// The boxes of the pool differ by +-10 % on the same binary (driver records r01 - r04: the latency-bound kernels stretch 25 - 30 % on a
// slow box, the issue-bound ones follow the clock the power budget allows), which hid every gain below that.  Three numbers that do
// not depend on the product's kernels place a box: the cost of a dependent launch boundary, the wave-instruction rate of independent
// v_fma_f32 streams at 4 waves per SIMD, and the shader clock that grid sustains (s_memtime ticks — one per shader cycle,
// MI355X_MICROARCH.md "s_memtime tick vs SQ PMC units" — against the constant 100 MHz s_memrealtime).  bench.py adds an HBM copy and a
// fixed 0.5 M-pair tile sort through surfel_debug_sort_pairs and prints `box_probe` + `ms_per_step_normalised`.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/surfel_debug.h"

namespace surfel {
int api_fail(int code, const char* what, hipError_t e);

namespace {

__global__ void probe_empty_kernel(int* p) {
    if (p && threadIdx.x == 999) *p = 0;
}

// 8 independent chains per lane: nothing for the scheduler to wait on but the VALU itself.  KIND 0: v_fma_f32 (VOP3, three VGPR sources),
// 1: v_add_f32 (VOP2), 2: v_pk_fma_f32 (two fp32 FMAs per lane and instruction)
template <int KIND>
__global__ void __launch_bounds__(256) probe_valu_kernel(unsigned long long* out, int iters) {
    float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f;
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = {a1, a0}, p5 = {a3, a2}, p6 = {a5, a4}, p7 = {a7, a6};
    const float c = 1.0001f, d = 1e-6f;
    const f2 pc = {c, c}, pd = {d, d};
    __syncthreads();
    const unsigned long long c0 = __builtin_readcyclecounter(), r0 = wall_clock64();
    for (int i = 0; i < iters; i++) {
        if (KIND == 0)
            asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                         "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(d));
        else if (KIND == 1)
            asm volatile("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n"
                         "v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(d));
        else
            asm volatile("v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n"
                         "v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9\n"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pc), "v"(pd));
    }
    const unsigned long long c1 = __builtin_readcyclecounter(), r1 = wall_clock64();
    if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p1.y + p2.x + p3.y + p4.x + p5.y + p6.x + p7.y == 12345.678f) out[0] = 1ull;      // (keeps the chains alive)
    if (threadIdx.x == 0) {
        out[4 * blockIdx.x + 4] = c1 - c0;      // shader cycles of this workgroup's loop
        out[4 * blockIdx.x + 5] = r1 - r0;      // the same span in 100 MHz ticks
        out[4 * blockIdx.x + 6] = r0;           // absolute: the grid's span = max r1 - min r0
        out[4 * blockIdx.x + 7] = r1;
    }
}

// A frozen stand-in for one visit of a blend-backward walk (the kernels this box is being compared for): six 16-B LDS reads, ~100 fp32
// multiply-adds on them, four transcendentals, two dozen compares / selects, and a 16-lane DPP reduction of 18 values (38 v_add_f32_dpp) —
// the instruction mix of profiles/r04_isa_walk_loops.md, at the kernels' own occupancy (4 workgroups per CU).  Never changed: its rate
// places a box for THIS kind of work better than a pure FMA stream does (boxes that differ by 30 % on v_fma_f32 differ by a few per
// cent on the blend kernels).
__global__ void __launch_bounds__(256, 4) probe_mix_kernel(unsigned long long* out, int iters) {
    __shared__ float4 s_rec[128 * 5];
    for (int k = threadIdx.x; k < 128 * 5; k += 256) s_rec[k] = make_float4(0.001f * k, 1.f + 0.002f * k, 0.5f - 0.0005f * k, 0.25f + 0.0001f * k);
    __syncthreads();
    const float px = (float)(threadIdx.x & 15), py = (float)(threadIdx.x >> 4);
    float T = 0.9f, X = 0.1f, acc = 0.f;
    const unsigned long long r0 = wall_clock64();
    for (int i = 0; i < iters; i++) {
        const int j = (i * 37 + (threadIdx.x >> 4)) & 127;
        const float4 q0 = s_rec[j * 5 + 0], q1 = s_rec[j * 5 + 1], q2 = s_rec[j * 5 + 2], q3 = s_rec[j * 5 + 3], q4 = s_rec[j * 5 + 4];
        const float4 qn = s_rec[((j + 1) & 127) * 5];
        // planes, intersection, alpha
        const float kx = __builtin_fmaf(px, q1.z, -q0.x), ky = __builtin_fmaf(px, q1.w, -q0.y), kz = __builtin_fmaf(px, q2.x, -q0.z);
        const float lx = __builtin_fmaf(py, q1.z, -q0.w), ly = __builtin_fmaf(py, q1.w, -q1.x), lz = __builtin_fmaf(py, q2.x, -q1.y);
        const float p0 = __builtin_fmaf(ky, lz, -(kz * ly)), p1 = __builtin_fmaf(kz, lx, -(kx * lz)), p2 = __builtin_fmaf(kx, ly, -(ky * lx));
        const float ip = __builtin_amdgcn_rcpf(p2 + 3.f);
        const float sx = p0 * ip, sy = p1 * ip;
        const float rho3 = __builtin_fmaf(sx, sx, sy * sy), dx = q2.y - px, dy = q2.z - py;
        const float rho2 = 2.f * __builtin_fmaf(dx, dx, dy * dy);
        const bool use3 = rho3 <= rho2;
        const float rho = fminf(rho3, rho2);
        const float depth = use3 ? __builtin_fmaf(sx, q1.z, sy * q1.w) + q2.x : q2.x;
        const float G = __expf(-0.5f * rho * 1e-3f);
        const float alpha = fminf(0.99f, q2.w * G * 0.01f);
        const bool ok = (p2 != 0.f) & (depth >= 0.2f) & (alpha >= 1.f / 255.f) & (j <= 100 + (i & 31));
        const float al = ok ? alpha : 0.f, dp = ok ? depth : 1.f;
        const float i1a = __builtin_amdgcn_rcpf(1.f - al);
        T = T * i1a; T = T > 4.f ? 0.9f : T;
        const float w = al * T, invd = __builtin_amdgcn_rcpf(dp);
        const float mm = __builtin_fmaf(-0.2004f, invd, 1.002f);
        float u = __builtin_fmaf(__builtin_fmaf(mm, __builtin_fmaf(mm, 0.7f, -0.6f), 0.3f), 1.5f, 0.2f);
        u = __builtin_fmaf(q3.w, 0.3f, u); u = __builtin_fmaf(q4.x, 0.2f, u); u = __builtin_fmaf(q4.y, 0.1f, u); u = __builtin_fmaf(dp, 0.05f, u);
        u = __builtin_fmaf(q3.x, 0.4f, u); u = __builtin_fmaf(q3.y, 0.5f, u); u = __builtin_fmaf(q3.z, 0.6f, u);
        u = ok ? u : 0.f;
        const float dLa = ok ? __builtin_fmaf(T, u, -(X * i1a)) : 0.f;
        X = __builtin_fmaf(w, u, X); X = X > 1e3f ? 0.1f : X;
        float dLz = __builtin_fmaf((2.f * w * 1.5f) * __builtin_fmaf(mm, 0.7f, -0.3f), 20.04f * invd * invd, w * 0.05f);
        dLz += (ok & (j == 17)) ? 0.3f : 0.f;
        const float nGG = -G * (q2.w * dLa);
        const float sxg = use3 ? sx : 0.f, syg = use3 ? sy : 0.f, ipg = use3 ? ip : 0.f, g2 = use3 ? 0.f : nGG * 2.f;
        const float ax = __builtin_fmaf(nGG, sxg, dLz * q1.z) * ipg, ay = __builtin_fmaf(nGG, syg, dLz * q1.w) * ipg;
        const float dp2 = -__builtin_fmaf(ax, sxg, ay * syg);
        float v[20];
        v[0] = __builtin_fmaf(ay, lz, -(dp2 * ly)); v[1] = __builtin_fmaf(dp2, lx, -(ax * lz)); v[2] = __builtin_fmaf(ax, ly, -(ay * lx));
        v[3] = __builtin_fmaf(ky, dp2, -(kz * ay)); v[4] = __builtin_fmaf(kz, ax, -(kx * dp2)); v[5] = __builtin_fmaf(kx, ay, -(ky * ax));
        v[6] = __builtin_fmaf(dLz, sxg, -__builtin_fmaf(px, v[0], py * v[3])); v[7] = __builtin_fmaf(dLz, syg, -__builtin_fmaf(px, v[1], py * v[4]));
        v[8] = dLz - __builtin_fmaf(px, v[2], py * v[5]); v[9] = g2 * dx; v[10] = g2 * dy;
        v[11] = w * 0.4f; v[12] = w * 0.5f; v[13] = w * 0.6f; v[14] = G * dLa; v[15] = w * 0.3f; v[16] = w * 0.2f; v[17] = w * 0.1f; v[18] = qn.x * 0.f; v[19] = 0.f;
        // 16-lane reduction of 18 values: 38 DPP adds (the rows walk's tree)
#pragma unroll
        for (int m = 0; m < 18; m += 2) {
            float o;
            asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %1, %1 row_ror:8 row_mask:0xf bank_mask:0xf\n\tv_add_f32_dpp %0, %2, %2 row_ror:8 row_mask:0xf bank_mask:0xc" : "=&v"(o) : "v"(v[m]), "v"(v[m + 1]));
            v[m] = o;
        }
#pragma unroll
        for (int m = 0; m < 20; m += 4) {
            float o;
            asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %1, %1 row_ror:12 row_mask:0xf bank_mask:0xf\n\tv_add_f32_dpp %0, %2, %2 row_ror:4 row_mask:0xf bank_mask:0xa" : "=&v"(o) : "v"(v[m]), "v"(v[(m + 2) % 20]));
            float t2, o2;
            asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %2, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\tv_add_f32_dpp %1, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf" : "=&v"(t2), "=&v"(o2) : "v"(o));
            acc += o2;
        }
    }
    const unsigned long long r1 = wall_clock64();
    if (acc + T + X == 12345.678f) out[0] = 1ull;
    if (threadIdx.x == 0) { out[4 * blockIdx.x + 6] = r0; out[4 * blockIdx.x + 7] = r1; }
}

// Dependent-load latency: one lane chases a permutation cycle through `n` words (stride chosen by the host so that every hop leaves the
// line and the page); ns per hop = the round trip of a load that misses L1 / L2 / MALL depending on the footprint.
__global__ void probe_chase_kernel(const uint32_t* __restrict__ next, int hops, unsigned long long* out) {
    if (threadIdx.x != 0) return;
    uint32_t i = 0;
    const unsigned long long r0 = wall_clock64();
    for (int k = 0; k < hops; k++) i = next[i];
    const unsigned long long r1 = wall_clock64();
    out[0] = (r1 - r0) + (i == 0xffffffffu ? 1ull : 0ull);
}
__global__ void probe_chase_init_kernel(uint32_t* __restrict__ next, uint32_t n, uint32_t stride) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) next[i] = (uint32_t)(((unsigned long long)i + stride) % n);
}

}  // namespace
}  // namespace surfel

using namespace surfel;

// ns per dependent global load through a cycle over `bytes` of `buf` (a power of two >= 1 MiB; stride 4099 words: co-prime, > a page)
extern "C" int surfel_debug_latency_probe(void* buf, int64_t bytes, int hops, float* ns_per_hop, void* stream) {
    if (!buf || bytes < (1 << 20) || hops <= 0 || !ns_per_hop) return api_fail(SURFEL_E_INVALID, "latency probe: bad arguments", hipSuccess);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const uint32_t n = (uint32_t)(bytes / 4) - 2;      // (the last two words receive the result)
    uint32_t* next = static_cast<uint32_t*>(buf);
    unsigned long long* out = reinterpret_cast<unsigned long long*>(next + (n & ~1u));
    hipLaunchKernelGGL(probe_chase_init_kernel, dim3((n + 255) / 256), dim3(256), 0, s, next, n & ~1u, 4099u | 1u);
    hipLaunchKernelGGL(probe_chase_kernel, dim3(1), dim3(64), 0, s, next, 256, out);      // warm code (not the data: the footprint decides what the hops hit)
    hipLaunchKernelGGL(probe_chase_kernel, dim3(1), dim3(64), 0, s, next, hops, out);
    unsigned long long ticks = 0;
    hipError_t e = hipStreamSynchronize(s);
    if (e == hipSuccess) e = hipMemcpy(&ticks, out, sizeof(ticks), hipMemcpyDeviceToHost);
    if (e != hipSuccess) return api_fail(SURFEL_E_HIP, "latency probe", e);
    *ns_per_hop = (float)((double)ticks * 10.0 / (double)hops);
    return 0;
}

extern "C" int surfel_debug_box_probe(void* scratch, int64_t scratch_bytes, float* out, void* stream) {
    // 1024 workgroups = 4 per CU = 4 waves per SIMD: all resident from the first cycle (8 per CU came out as two dispatch rounds on some
    // boxes, which measures the dispatcher), and four waves of independent FMAs saturate a SIMD's fp32 pipe
    constexpr int kLaunches = 256, kBlocks = 1024, kIters = 16384;
    if (!scratch || !out || scratch_bytes < (int64_t)((4 * kBlocks + 4) * sizeof(unsigned long long))) return api_fail(SURFEL_E_INVALID, "box probe: scratch too small", hipSuccess);
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipEvent_t e0, e1, e2, e3;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess || hipEventCreate(&e2) != hipSuccess || hipEventCreate(&e3) != hipSuccess)
        return api_fail(SURFEL_E_HIP, "box probe: event creation failed", hipGetLastError());
    unsigned long long* buf = static_cast<unsigned long long*>(scratch);
    for (int i = 0; i < 16; i++) hipLaunchKernelGGL(probe_empty_kernel, dim3(1), dim3(64), 0, s, (int*)nullptr);      // warm code, queue
    (void)hipEventRecord(e0, s);
    for (int i = 0; i < kLaunches; i++) hipLaunchKernelGGL(probe_empty_kernel, dim3(1), dim3(64), 0, s, (int*)nullptr);
    (void)hipEventRecord(e1, s);
    hipLaunchKernelGGL(probe_valu_kernel<0>, dim3(kBlocks), dim3(256), 0, s, buf, 64);      // warm
    (void)hipEventRecord(e2, s);
    hipLaunchKernelGGL(probe_valu_kernel<0>, dim3(kBlocks), dim3(256), 0, s, buf, kIters);
    (void)hipEventRecord(e3, s);
    hipError_t e = hipEventSynchronize(e3);
    float ms_launch = 0.f, ms_valu = 0.f;
    if (e == hipSuccess) e = hipEventElapsedTime(&ms_launch, e0, e1);
    if (e == hipSuccess) e = hipEventElapsedTime(&ms_valu, e2, e3);
    static unsigned long long host[4 * kBlocks + 4];
    if (e == hipSuccess) e = hipMemcpy(host, buf, sizeof(host), hipMemcpyDeviceToHost);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); (void)hipEventDestroy(e2); (void)hipEventDestroy(e3);
    if (e != hipSuccess) return api_fail(SURFEL_E_HIP, "box probe", e);
    double cyc = 0.0, ticks = 0.0;
    unsigned long long first = ~0ull, last = 0ull;
    for (int b = 0; b < kBlocks; b++) {
        cyc += (double)host[4 * b + 4]; ticks += (double)host[4 * b + 5];
        if (host[4 * b + 6] < first) first = host[4 * b + 6];
        if (host[4 * b + 7] > last) last = host[4 * b + 7];
    }
    const double insts = (double)kBlocks * 4.0 * (double)kIters * 8.0;      // wave-instructions of the grid
    const double clk = ticks > 0.0 ? cyc / ticks * 0.1 : 0.0;               // GHz
    const double span_ns = (double)(last - first) * 10.0;                   // first loop entry -> last loop exit, on the device's own 100 MHz clock
    out[0] = 1e3f * ms_launch / (float)kLaunches;                           // us per dependent launch boundary
    out[1] = (float)(insts / ((double)ms_valu * 1e-3) / 1e9);               // G wave-instructions / s, whole chip, launch to end (events)
    out[2] = (float)clk;                                                    // shader clock under that grid, GHz
    out[3] = ms_valu;
    // shader cycles per wave-instruction per SIMD over the grid's own span (1024 SIMDs; the nominal figure is 2: SIMD-32 fp32)
    out[4] = span_ns > 0.0 ? (float)(span_ns * clk * 1024.0 / insts) : 0.f;
    out[5] = span_ns > 0.0 ? (float)(insts / span_ns) : 0.f;               // G wave-instructions / s over that span (no launch ramp)
    // the same grid of v_add_f32 (VOP2) and of v_pk_fma_f32: shader cycles per wave-instruction per SIMD over the grid's own span
    for (int kind = 1; kind <= 2; kind++) {
        if (kind == 1) hipLaunchKernelGGL(probe_valu_kernel<1>, dim3(kBlocks), dim3(256), 0, s, buf, kIters / 4);
        else hipLaunchKernelGGL(probe_valu_kernel<2>, dim3(kBlocks), dim3(256), 0, s, buf, kIters / 4);
        e = hipStreamSynchronize(s);
        if (e == hipSuccess) e = hipMemcpy(host, buf, sizeof(host), hipMemcpyDeviceToHost);
        if (e != hipSuccess) return api_fail(SURFEL_E_HIP, "box probe", e);
        double cy = 0.0, tk = 0.0;
        unsigned long long f0 = ~0ull, l0 = 0ull;
        for (int b = 0; b < kBlocks; b++) {
            cy += (double)host[4 * b + 4]; tk += (double)host[4 * b + 5];
            if (host[4 * b + 6] < f0) f0 = host[4 * b + 6];
            if (host[4 * b + 7] > l0) l0 = host[4 * b + 7];
        }
        const double ck = tk > 0.0 ? cy / tk * 0.1 : 0.0, sp = (double)(l0 - f0) * 10.0, ins = (double)kBlocks * 4.0 * (double)(kIters / 4) * 8.0;
        out[5 + kind] = sp > 0.0 ? (float)(sp * ck * 1024.0 / ins) : 0.f;
    }
    {   // the blend-like mix: M wave-visits per second over the grid's own span
        constexpr int kMixIters = 2048;
        hipLaunchKernelGGL(probe_mix_kernel, dim3(kBlocks), dim3(256), 0, s, buf, 64);
        hipLaunchKernelGGL(probe_mix_kernel, dim3(kBlocks), dim3(256), 0, s, buf, kMixIters);
        e = hipStreamSynchronize(s);
        if (e == hipSuccess) e = hipMemcpy(host, buf, sizeof(host), hipMemcpyDeviceToHost);
        if (e != hipSuccess) return api_fail(SURFEL_E_HIP, "box probe", e);
        unsigned long long f0 = ~0ull, l0 = 0ull;
        for (int b = 0; b < kBlocks; b++) {
            if (host[4 * b + 6] < f0) f0 = host[4 * b + 6];
            if (host[4 * b + 7] > l0) l0 = host[4 * b + 7];
        }
        const double sp = (double)(l0 - f0) * 10.0;
        out[8] = sp > 0.0 ? (float)((double)kBlocks * 4.0 * (double)kMixIters / sp * 1e3) : 0.f;      // M wave-visits / s
    }
    return 0;
}
