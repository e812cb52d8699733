// Microbenchmark: what a global atomic add costs on MI355X depending on its scope, and where it executes.
//   agent scope (HIP's atomicAdd)      : coherent across the 8 XCDs -> performed memory-side, over the fabric
//   workgroup scope on a per-XCD copy  : no sc1 bit -> performed in the issuing XCD's own L2 (all global atomics execute at L2);
//                                        correct as long as every XCD only touches its own copy (XCC_ID hardware register)
// Pattern: the counting-binning one — 480 k instances over 2 500 tile counters (8-B stride), uniform and skewed.
// Build: hipcc --offload-arch=gfx950 -O3 -o atomic_scope atomic_scope.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

__device__ __forceinline__ uint32_t xcc_id() { return __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)); }      // HW_REG_XCC_ID[3:0]

template <int SCOPE /*0 agent, 1 workgroup on the XCD's copy*/, bool RET>
__global__ void __launch_bounds__(256) k_atomic(uint32_t* counters, const uint32_t* tile_of, int n, int ntiles, uint32_t* out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint32_t t = tile_of[i];
    uint32_t r;
    if (SCOPE == 0) r = __hip_atomic_fetch_add(counters + 2 * (size_t)t, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else r = __hip_atomic_fetch_add(counters + 2 * ((size_t)xcc_id() * ntiles + t), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (RET) out[i] = r;
}

__global__ void k_xcc(uint32_t* where) { if (threadIdx.x == 0) where[blockIdx.x] = xcc_id(); }

template <typename F>
static float time_us(F f, int reps) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    f(); hipDeviceSynchronize();
    hipEventRecord(a);
    for (int r = 0; r < reps; r++) f();
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms * 1e3f / reps;
}

int main() {
    const int n = 480000, ntiles = 2500, reps = 50;
    for (int skew = 0; skew < 2; skew++) {
        std::vector<uint32_t> h(n);
        srand(7);
        for (int i = 0; i < n; i++) h[i] = (skew && (rand() % 10) < 3) ? (uint32_t)(rand() % 10) : (uint32_t)(rand() % ntiles);
        uint32_t *tile_of, *cnt, *out;
        hipMalloc(&tile_of, n * 4); hipMalloc(&cnt, 8 * ntiles * 8 + 256); hipMalloc(&out, n * 4);
        hipMemcpy(tile_of, h.data(), n * 4, hipMemcpyHostToDevice);
        const dim3 g((n + 255) / 256), b(256);
        auto zero = [&] { hipMemsetAsync(cnt, 0, 8 * ntiles * 8, 0); };
        const float t_zero = time_us(zero, reps);
        const float t0 = time_us([&] { zero(); hipLaunchKernelGGL((k_atomic<0, false>), g, b, 0, 0, cnt, tile_of, n, ntiles, out); }, reps);
        const float t1 = time_us([&] { zero(); hipLaunchKernelGGL((k_atomic<0, true>), g, b, 0, 0, cnt, tile_of, n, ntiles, out); }, reps);
        const float t2 = time_us([&] { zero(); hipLaunchKernelGGL((k_atomic<1, false>), g, b, 0, 0, cnt, tile_of, n, ntiles, out); }, reps);
        const float t3 = time_us([&] { zero(); hipLaunchKernelGGL((k_atomic<1, true>), g, b, 0, 0, cnt, tile_of, n, ntiles, out); }, reps);
        // correctness of the per-XCD copies: their sum must be the histogram
        zero(); hipLaunchKernelGGL((k_atomic<1, false>), g, b, 0, 0, cnt, tile_of, n, ntiles, out); hipDeviceSynchronize();
        std::vector<uint32_t> c(8 * ntiles * 2), ref(ntiles, 0);
        hipMemcpy(c.data(), cnt, c.size() * 4, hipMemcpyDeviceToHost);
        for (int i = 0; i < n; i++) ref[h[i]]++;
        int bad = 0; long long per_xcd[8] = {0};
        for (int t = 0; t < ntiles; t++) {
            uint32_t s = 0;
            for (int x = 0; x < 8; x++) { s += c[2 * ((size_t)x * ntiles + t)]; per_xcd[x] += c[2 * ((size_t)x * ntiles + t)]; }
            bad += s != ref[t];
        }
        printf("%s: memset %.1f us | agent no-return %.1f, agent returning %.1f | XCD-local no-return %.1f, XCD-local returning %.1f us (memset included)"
               " | per-XCD copies sum to the histogram: %s (per XCD:", skew ? "skewed (30 %% on 10 tiles)" : "uniform", t_zero, t0, t1, t2, t3, bad ? "NO" : "yes");
        for (int x = 0; x < 8; x++) printf(" %lld", per_xcd[x]);
        printf(")\n");
        hipFree(tile_of); hipFree(cnt); hipFree(out);
    }
    uint32_t* where; hipMalloc(&where, 4096 * 4);
    hipLaunchKernelGGL(k_xcc, dim3(4096), dim3(64), 0, 0, where); hipDeviceSynchronize();
    std::vector<uint32_t> w(4096); hipMemcpy(w.data(), where, 4096 * 4, hipMemcpyDeviceToHost);
    int match = 0; for (int i = 0; i < 4096; i++) match += (int)w[i] == i % 8;
    printf("XCC_ID == blockIdx %% 8 for %d of 4096 workgroups; first 16:", match);
    for (int i = 0; i < 16; i++) printf(" %u", w[i]);
    printf("\n");
    return 0;
}
