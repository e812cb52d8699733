// box_probe.hip — what THIS GPU box sustains, measured in the bench process right before the timed window (gfx950).
//
// The boxes of the pool differ by +-10 % on the same binary (driver records r01 - r04: the latency-bound kernels stretch 25 - 30 % on a
// slow box, the issue-bound ones follow the clock the power budget allows), which hid every gain below that.  Three numbers that do
// not depend on the product's kernels place a box: the cost of a dependent launch boundary, the wave-instruction rate of independent
// v_fma_f32 streams at 8 waves per SIMD, and the shader clock that grid sustains (s_memtime ticks — one per shader cycle,
// MI355X_MICROARCH.md "s_memtime tick vs SQ PMC units" — against the constant 100 MHz s_memrealtime).  bench.py adds an HBM copy and a
// fixed 0.5 M-pair tile sort through surfel_debug_sort_pairs and prints `box_probe` + `ms_per_step_normalised`.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/surfel_hip.h"

namespace surfel {
int api_fail(int code, const char* what, hipError_t e);

namespace {

__global__ void probe_empty_kernel(int* p) {
    if (p && threadIdx.x == 999) *p = 0;
}

// 8 independent v_fma_f32 chains per lane: nothing for the scheduler to wait on but the VALU itself
__global__ void __launch_bounds__(256) probe_valu_kernel(unsigned long long* out, int iters) {
    float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f;
    const float c = 1.0001f, d = 1e-6f;
    __syncthreads();
    const unsigned long long c0 = __builtin_readcyclecounter(), r0 = wall_clock64();
    for (int i = 0; i < iters; i++) {
        asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                     "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(d));
    }
    const unsigned long long c1 = __builtin_readcyclecounter(), r1 = wall_clock64();
    if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 12345.678f) out[0] = 1ull;      // (keeps the chains alive)
    if (threadIdx.x == 0) {
        out[4 * blockIdx.x + 4] = c1 - c0;      // shader cycles of this workgroup's loop
        out[4 * blockIdx.x + 5] = r1 - r0;      // the same span in 100 MHz ticks
        out[4 * blockIdx.x + 6] = r0;           // absolute: the grid's span = max r1 - min r0
        out[4 * blockIdx.x + 7] = r1;
    }
}

}  // namespace
}  // namespace surfel

using namespace surfel;

extern "C" int surfel_debug_box_probe(void* scratch, int64_t scratch_bytes, float* out, void* stream) {
    constexpr int kLaunches = 256, kBlocks = 2048 /* 256 CUs x 4 SIMDs x 8 waves / 4 waves per block */, kIters = 16384;
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
    hipLaunchKernelGGL(probe_valu_kernel, dim3(kBlocks), dim3(256), 0, s, buf, 64);      // warm
    (void)hipEventRecord(e2, s);
    hipLaunchKernelGGL(probe_valu_kernel, dim3(kBlocks), dim3(256), 0, s, buf, kIters);
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
    return 0;
}
