// loss_trace.hip — where the fused training-loss launches spend their time (gfx950): a stand-alone harness around the PRODUCT's kernel
// bodies (csrc/train_loss_body.h, train_post_body.h — included, not copied) that stamps every workgroup's start and end on the 100 MHz
// clock and records the CU it ran on.  No library, no torch:
//     hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=fast -I 2d-gaussian-splatting_amd/csrc scripts/loss_trace.hip -o /tmp/loss_trace && /tmp/loss_trace [H W]
// Prints, for the forward and the backward launch at H x W (default 800 x 800): event-timed kernel time without stamps, the span of the
// stamped run, workgroup durations (SSIM / post-processing bodies), how many workgroups were resident over time, and the time-to-first
// / tail of the grid — the numbers that say whether the launch is bound by dispatch, by residency (LDS), by the bodies' own latency
// chain or by a tail.  PAD=<bytes> (environment) adds LDS per workgroup to lower the residency.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <vector>

#include "surfel_common.h"
#include "train_loss_body.h"
#include "train_post_body.h"

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

using namespace surfel;
#ifndef MINW_F
#define MINW_F 1      // -DMINW_F=n / -DMINW_B=n: __launch_bounds__(256, n) — the register budget of n waves per SIMD (the product: no bound)
#endif
#ifndef MINW_B
#define MINW_B 1
#endif
constexpr int SR11 = 5;
constexpr size_t cmax(size_t a, size_t b) { return a > b ? a : b; }

__device__ __forceinline__ uint32_t hw_id() {
    uint32_t v, x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(v));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    return ((x & 0xfu) << 16) | ((v >> 8) & 0xfu) | (((v >> 13) & 0x7u) << 4) | (((v >> 12) & 0x1u) << 8);      // xcc | cu_id | se_id | sh_id
}

template <bool TRACE>
__global__ __launch_bounds__(256, MINW_F) void fwd_kernel(int n_ssim, int n_ssim_pad, int n_post, int H, int W, const float* __restrict__ img, const float* __restrict__ gt,
                                                  float* __restrict__ dmaps, size_t map_stride, float* __restrict__ partials, lossk::SsimWin win,
                                                  const float* __restrict__ allmap, const float* __restrict__ cam, float ratio, float* __restrict__ maps,
                                                  float* __restrict__ post_partials, unsigned long long* trace) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    unsigned long long t0 = 0;
    if (TRACE) t0 = wall_clock64();
    const int b = blockIdx.x;
    if (b < n_ssim_pad) {
        if (b < n_ssim) lossk::ssim_fwd_body<SR11>(smem, b, n_ssim, H, W, img, gt, dmaps, map_stride, partials, win);
    } else {
        postk::post_fwd_body(smem, b - n_ssim_pad, n_post, H, W, allmap, cam, ratio, maps, post_partials);
    }
    if (TRACE) {
        __syncthreads();
        if (threadIdx.x == 0) { trace[3 * b] = t0; trace[3 * b + 1] = wall_clock64(); trace[3 * b + 2] = hw_id(); }
    }
}

template <bool TRACE>
__global__ __launch_bounds__(256, MINW_B) void bwd_kernel(int n_ssim, int n_ssim_pad, int n_post, int H, int W, const float* __restrict__ img, const float* __restrict__ gt,
                                                  const float* __restrict__ dmaps, size_t map_stride, float c_l1, float c_ssim, const float* __restrict__ g_dev,
                                                  float* __restrict__ grad_img, lossk::SsimWin win, const float* __restrict__ allmap, const float* __restrict__ cam,
                                                  float ratio, const float* __restrict__ gmaps, float c_normal, float c_dist, float* __restrict__ gall,
                                                  unsigned long long* trace) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    unsigned long long t0 = 0;
    if (TRACE) t0 = wall_clock64();
    const int b = blockIdx.x;
    if (b < n_ssim_pad) {
        if (b < n_ssim) lossk::ssim_bwd_body<SR11>(smem, b, n_ssim, H, W, img, gt, dmaps, map_stride, c_l1, c_ssim, g_dev, g_dev, grad_img, win);
    } else {
        postk::post_bwd_body(smem, b - n_ssim_pad, n_post, H, W, allmap, cam, ratio, gmaps, c_normal, c_dist, g_dev, gall);
    }
    if (TRACE) {
        __syncthreads();
        if (threadIdx.x == 0) { trace[3 * b] = t0; trace[3 * b + 1] = wall_clock64(); trace[3 * b + 2] = hw_id(); }
    }
}

static void fill(std::vector<float>& v, unsigned seed, float lo, float hi) {
    unsigned s = seed * 2654435761u + 12345u;
    for (auto& x : v) { s = s * 1664525u + 1013904223u; x = lo + (hi - lo) * (float)(s >> 8) * (1.f / 16777216.f); }
}

static void report(const char* name, const std::vector<unsigned long long>& tr, int n_ssim, int n_pad, int n_post) {
    const int nb = n_pad + n_post;
    unsigned long long first = ~0ull, last = 0ull;
    for (int b = 0; b < nb; b++) {
        if (b >= n_ssim && b < n_pad) continue;
        first = std::min(first, tr[3 * b]); last = std::max(last, tr[3 * b + 1]);
    }
    auto stats = [&](int lo, int hi, const char* what) {
        std::vector<double> d;
        for (int b = lo; b < hi; b++) d.push_back((double)(tr[3 * b + 1] - tr[3 * b]) * 0.01);
        if (d.empty()) return;
        std::sort(d.begin(), d.end());
        double sum = 0; for (double x : d) sum += x;
        printf("  %-5s %5zu workgroups: duration us min %.2f  p10 %.2f  median %.2f  p90 %.2f  max %.2f  (sum %.0f us = %.1f resident on average over the span)\n", what, d.size(),
               d.front(), d[d.size() / 10], d[d.size() / 2], d[d.size() * 9 / 10], d.back(), sum, sum / ((double)(last - first) * 0.01));
    };
    printf("%s: span first start -> last end %.2f us\n", name, (double)(last - first) * 0.01);
    stats(0, n_ssim, "ssim");
    stats(n_pad, nb, "post");
    // residency over time (1-us bins) and starts per bin
    const int bins = (int)((last - first) / 100) + 1;
    std::vector<int> res(bins, 0), starts(bins, 0), res_s(bins, 0);
    for (int b = 0; b < nb; b++) {
        if (b >= n_ssim && b < n_pad) continue;
        const int s = (int)((tr[3 * b] - first) / 100), e = (int)((tr[3 * b + 1] - first) / 100);
        starts[s]++;
        for (int k = s; k <= e && k < bins; k++) { res[k]++; if (b < n_ssim) res_s[k]++; }
    }
    printf("  per-us bin: resident workgroups (of which ssim) / started:");
    for (int k = 0; k < bins; k++) printf(" %d(%d)/%d", res[k], res_s[k], starts[k]);
    printf("\n");
    // per-CU: workgroups served and the largest number resident at once
    int cus = 0, maxres = 0;
    std::vector<int> served;
    {
        std::vector<unsigned> ids;
        for (int b = 0; b < nb; b++) if (!(b >= n_ssim && b < n_pad)) ids.push_back((unsigned)tr[3 * b + 2]);
        std::sort(ids.begin(), ids.end());
        ids.erase(std::unique(ids.begin(), ids.end()), ids.end());
        cus = (int)ids.size();
        for (unsigned id : ids) {
            std::vector<std::pair<unsigned long long, int>> e;
            int n = 0;
            for (int b = 0; b < nb; b++) {
                if (b >= n_ssim && b < n_pad) continue;
                if ((unsigned)tr[3 * b + 2] == id) { e.push_back({tr[3 * b], 1}); e.push_back({tr[3 * b + 1], -1}); n++; }
            }
            std::sort(e.begin(), e.end());
            int cur = 0, mx = 0;
            for (auto& p : e) { cur += p.second; mx = std::max(mx, cur); }
            maxres = std::max(maxres, mx);
            served.push_back(n);
        }
    }
    std::sort(served.begin(), served.end());
    printf("  %d CUs seen; workgroups per CU min %d median %d max %d; most resident at once on one CU: %d\n", cus, served.front(), served[served.size() / 2], served.back(), maxres);
}

int main(int argc, char** argv) {
    const int H = argc > 2 ? atoi(argv[1]) : 800, W = argc > 2 ? atoi(argv[2]) : 800;
    const int pad = getenv("PAD") ? atoi(getenv("PAD")) : 0;
    const size_t HW = (size_t)H * W;
    const int n_ssim = ((W + lossk::ST - 1) / lossk::ST) * ((H + lossk::ST - 1) / lossk::ST) * 3, n_pad = (n_ssim + 7) / 8 * 8;
    const int n_post = ((W + postk::PT - 1) / postk::PT) * ((H + postk::PT - 1) / postk::PT), nb = n_pad + n_post;
    const size_t lds_f = cmax(lossk::ssim_fwd_lds<SR11>(), postk::post_fwd_lds()) + pad, lds_b = cmax(lossk::ssim_bwd_lds<SR11>(), postk::post_bwd_lds()) + pad;
    printf("# %d x %d: %d ssim + %d post workgroups, LDS fwd %zu B (%d / CU), bwd %zu B (%d / CU)\n", H, W, n_ssim, n_post, lds_f, (int)(160 * 1024 / lds_f), lds_b, (int)(160 * 1024 / lds_b));
    std::vector<float> h_img(3 * HW), h_gt(3 * HW), h_all(10 * HW), h_cam(21);
    fill(h_img, 1, 0.f, 1.f); fill(h_gt, 2, 0.f, 1.f); fill(h_all, 3, 0.2f, 1.f); fill(h_cam, 4, -1.f, 1.f);
    float *img, *gt, *dmaps, *part, *allmap, *cam, *ppart, *gimg, *gall, *one;
    unsigned long long* trace;
    CHECK(hipMalloc(&img, 3 * HW * 4)); CHECK(hipMalloc(&gt, 3 * HW * 4)); CHECK(hipMalloc(&dmaps, 9 * HW * 4)); CHECK(hipMalloc(&part, 2 * n_ssim * 4 + 64));
    CHECK(hipMalloc(&allmap, 10 * HW * 4)); CHECK(hipMalloc(&cam, 21 * 4)); CHECK(hipMalloc(&ppart, 2 * n_post * 4 + 64)); CHECK(hipMalloc(&gimg, 3 * HW * 4));
    CHECK(hipMalloc(&gall, 10 * HW * 4)); CHECK(hipMalloc(&one, 4)); CHECK(hipMalloc(&trace, 3 * nb * 8));
    CHECK(hipMemcpy(img, h_img.data(), 3 * HW * 4, hipMemcpyHostToDevice)); CHECK(hipMemcpy(gt, h_gt.data(), 3 * HW * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(allmap, h_all.data(), 10 * HW * 4, hipMemcpyHostToDevice)); CHECK(hipMemcpy(cam, h_cam.data(), 21 * 4, hipMemcpyHostToDevice));
    const float onef = 1.f;
    CHECK(hipMemcpy(one, &onef, 4, hipMemcpyHostToDevice));
    lossk::SsimWin win;
    for (int i = 0; i < 15; i++) win.w[i] = i < 11 ? lossk::kG11[i] : 0.f;
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(fwd_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_f));
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(fwd_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_f));
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(bwd_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_b));
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(bwd_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_b));
    auto fwd = [&](bool tr) {
        if (tr) hipLaunchKernelGGL(fwd_kernel<true>, dim3(nb), dim3(256), lds_f, 0, n_ssim, n_pad, n_post, H, W, img, gt, dmaps, 3 * HW, part, win, allmap, cam, 1.f, (float*)nullptr, ppart, trace);
        else hipLaunchKernelGGL(fwd_kernel<false>, dim3(nb), dim3(256), lds_f, 0, n_ssim, n_pad, n_post, H, W, img, gt, dmaps, 3 * HW, part, win, allmap, cam, 1.f, (float*)nullptr, ppart, trace);
    };
    auto bwd = [&](bool tr) {
        if (tr) hipLaunchKernelGGL(bwd_kernel<true>, dim3(nb), dim3(256), lds_b, 0, n_ssim, n_pad, n_post, H, W, img, gt, dmaps, 3 * HW, 0.8f / (3 * HW), -0.2f / (3 * HW), one, gimg, win, allmap, cam, 1.f, (const float*)nullptr, 0.05f / HW, 1000.f / HW, gall, trace);
        else hipLaunchKernelGGL(bwd_kernel<false>, dim3(nb), dim3(256), lds_b, 0, n_ssim, n_pad, n_post, H, W, img, gt, dmaps, 3 * HW, 0.8f / (3 * HW), -0.2f / (3 * HW), one, gimg, win, allmap, cam, 1.f, (const float*)nullptr, 0.05f / HW, 1000.f / HW, gall, trace);
    };
    hipEvent_t e0, e1, e2;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1)); CHECK(hipEventCreate(&e2));
    for (int i = 0; i < 10; i++) { fwd(false); bwd(false); }
    CHECK(hipDeviceSynchronize());
    float best_f = 1e9f, best_b = 1e9f;
    for (int rep = 0; rep < 5; rep++) {
        CHECK(hipEventRecord(e0, 0));
        for (int i = 0; i < 20; i++) fwd(false);
        CHECK(hipEventRecord(e1, 0));
        for (int i = 0; i < 20; i++) bwd(false);
        CHECK(hipEventRecord(e2, 0));
        CHECK(hipEventSynchronize(e2));
        float a, b;
        CHECK(hipEventElapsedTime(&a, e0, e1)); CHECK(hipEventElapsedTime(&b, e1, e2));
        best_f = std::min(best_f, a / 20.f); best_b = std::min(best_b, b / 20.f);
    }
    printf("back-to-back launches, no stamps: forward %.2f us, backward %.2f us per launch\n", best_f * 1e3f, best_b * 1e3f);
    std::vector<unsigned long long> tr(3 * nb);
    for (int rep = 0; rep < 2; rep++) {
        fwd(true);
        CHECK(hipDeviceSynchronize());
        CHECK(hipMemcpy(tr.data(), trace, 3 * nb * 8, hipMemcpyDeviceToHost));
        if (rep == 1) report("forward", tr, n_ssim, n_pad, n_post);
        bwd(true);
        CHECK(hipDeviceSynchronize());
        CHECK(hipMemcpy(tr.data(), trace, 3 * nb * 8, hipMemcpyDeviceToHost));
        if (rep == 1) report("backward", tr, n_ssim, n_pad, n_post);
    }
    {   // FNV-1a of every output: two builds of the bodies must agree to the bit
        auto hash = [&](const void* dev, size_t bytes) {
            std::vector<unsigned char> h(bytes);
            if (hipMemcpy(h.data(), dev, bytes, hipMemcpyDeviceToHost) != hipSuccess) return 0ull;
            unsigned long long x = 1469598103934665603ull;
            for (unsigned char c : h) { x ^= c; x *= 1099511628211ull; }
            return x;
        };
        if (getenv("DUMP")) {      // grad_img as raw floats, for a numeric comparison of two builds
            std::vector<float> h(3 * HW);
            if (hipMemcpy(h.data(), gimg, 3 * HW * 4, hipMemcpyDeviceToHost) == hipSuccess) { FILE* f = fopen(getenv("DUMP"), "wb"); if (f) { fwrite(h.data(), 4, h.size(), f); fclose(f); } }
        }
        printf("output hashes: dmaps %016llx partials %016llx post_partials %016llx grad_img %016llx gall %016llx\n", hash(dmaps, 9 * HW * 4), hash(part, 2 * n_ssim * 4),
               hash(ppart, 2 * n_post * 4), hash(gimg, 3 * HW * 4), hash(gall, 7 * HW * 4));
    }
    return 0;
}
