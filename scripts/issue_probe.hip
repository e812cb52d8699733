// issue_probe.hip — how fast ONE SIMD of this box issues each kind of VALU instruction the blend walks are made of (gfx950).
// Stand-alone (no library, no torch):   hipcc --offload-arch=gfx950 -O3 scripts/issue_probe.hip -o /tmp/issue_probe && /tmp/issue_probe
// Every kernel: 256 workgroups of W x 4 waves, each claiming 100 KB of LDS so that exactly ONE workgroup sits on every CU (W waves per
// SIMD, all co-resident: default W = 4), 8 independent chains per lane, so that nothing but the VALU's own issue rate limits the loop.
// Printed per kind: shader cycles (s_memtime) per wave-instruction per SIMD = a wave's loop cycles / instructions / W (mean over the
// waves; max in brackets), and the chip-wide G wave-instructions/s over the grid's span (100 MHz clock).  `issue_probe ITERS W`:
// W = 1 turns the 8 chains of one wave into the only work of a SIMD (latency-bound issue of one wave).
// The question behind it (round 5): box_probe measured v_add_f32 at 3.7 cycles against v_fma_f32 at 2.9 — is that the opcode, the
// encoding (VOP2 / VOP3), the operands' register banks, or the DPP path?  The rows walk spends 38 of its 190 VALU instructions per
// visit in v_add_f32_dpp; the answer decides whether they should be v_fmac_f32_dpp with a 1.0 multiplier (same bits).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

constexpr int kBlocks = 256, kMaxWaves = 16;

#define ACC "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
#define REP8(P, S) P "0" S "\n" P "1" S "\n" P "2" S "\n" P "3" S "\n" P "4" S "\n" P "5" S "\n" P "6" S "\n" P "7" S "\n"

// the asm statement is repeated 4 x per loop trip (3 scalar loop instructions per 32 vector ones); NINST = wave-instructions of one copy
#define PROBE(NAME, NINST, ...)                                                                                               \
    __global__ void __launch_bounds__(1024) NAME(unsigned long long* out, int iters, float c, float d) {                      \
        float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, \
              a7 = a0 + 7.f;                                                                                                  \
        __syncthreads();                                                                                                      \
        const unsigned long long c0 = __builtin_readcyclecounter(), r0 = wall_clock64();                                      \
        for (int i = 0; i < iters; i++) { __VA_ARGS__; __VA_ARGS__; __VA_ARGS__; __VA_ARGS__; }                               \
        const unsigned long long c1 = __builtin_readcyclecounter(), r1 = wall_clock64();                                      \
        if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 12345.678f) out[0] = 1ull;                                               \
        extern __shared__ float lds_claim[];                                                                                  \
        if (iters < 0) lds_claim[threadIdx.x] = a0;                                                                           \
        if ((threadIdx.x & 63) == 0) {                                                                                        \
            const int w = 4 + 4 * (blockIdx.x * kMaxWaves + (threadIdx.x >> 6));                                              \
            out[w] = c1 - c0; out[w + 1] = r1 - r0; out[w + 2] = r0; out[w + 3] = r1;                                         \
        }                                                                                                                     \
    }                                                                                                                         \
    constexpr int NAME##_n = 4 * (NINST);

#define DPP8 " row_ror:8 row_mask:0xf bank_mask:0xf"
#define DPPC " row_ror:8 row_mask:0xf bank_mask:0xc"
#define QP " quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"
#define SHR1 " row_shr:1 row_mask:0xf bank_mask:0xf"

PROBE(k_fma, 8, asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                                  "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n" : ACC : "v"(c), "v"(d)))
PROBE(k_fma_one, 8, asm volatile("v_fma_f32 %0, %0, 1.0, %8\n v_fma_f32 %1, %1, 1.0, %8\n v_fma_f32 %2, %2, 1.0, %8\n v_fma_f32 %3, %3, 1.0, %8\n"
                                 "v_fma_f32 %4, %4, 1.0, %8\n v_fma_f32 %5, %5, 1.0, %8\n v_fma_f32 %6, %6, 1.0, %8\n v_fma_f32 %7, %7, 1.0, %8\n" : ACC : "v"(d)))
PROBE(k_add_inl, 8, asm volatile("v_add_f32 %0, 1.0, %0\n v_add_f32 %1, 1.0, %1\n v_add_f32 %2, 1.0, %2\n v_add_f32 %3, 1.0, %3\n"
                                 "v_add_f32 %4, 1.0, %4\n v_add_f32 %5, 1.0, %5\n v_add_f32 %6, 1.0, %6\n v_add_f32 %7, 1.0, %7\n" : ACC))
PROBE(k_sub, 8, asm volatile("v_sub_f32 %0, %0, %8\n v_sub_f32 %1, %1, %8\n v_sub_f32 %2, %2, %8\n v_sub_f32 %3, %3, %8\n"
                             "v_sub_f32 %4, %4, %8\n v_sub_f32 %5, %5, %8\n v_sub_f32 %6, %6, %8\n v_sub_f32 %7, %7, %8\n" : ACC : "v"(d)))
PROBE(k_fma_3src, 8, asm volatile("v_fma_f32 %0, %1, %2, %0\n v_fma_f32 %1, %2, %3, %1\n v_fma_f32 %2, %3, %4, %2\n v_fma_f32 %3, %4, %5, %3\n"
                                  "v_fma_f32 %4, %5, %6, %4\n v_fma_f32 %5, %6, %7, %5\n v_fma_f32 %6, %7, %8, %6\n v_fma_f32 %7, %8, %8, %7\n" : ACC : "v"(d)))
PROBE(k_fmac, 8, asm volatile(REP8("v_fmac_f32 %", ", %8, %9") : ACC : "v"(c), "v"(d)))
PROBE(k_add, 8, asm volatile("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n"
                             "v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8\n" : ACC : "v"(d)))
PROBE(k_add_rev, 8, asm volatile("v_add_f32 %0, %8, %0\n v_add_f32 %1, %8, %1\n v_add_f32 %2, %8, %2\n v_add_f32 %3, %8, %3\n"
                                 "v_add_f32 %4, %8, %4\n v_add_f32 %5, %8, %5\n v_add_f32 %6, %8, %6\n v_add_f32 %7, %8, %7\n" : ACC : "v"(d)))
PROBE(k_add_e64, 8, asm volatile("v_add_f32_e64 %0, %0, %8\n v_add_f32_e64 %1, %1, %8\n v_add_f32_e64 %2, %2, %8\n v_add_f32_e64 %3, %3, %8\n"
                                 "v_add_f32_e64 %4, %4, %8\n v_add_f32_e64 %5, %5, %8\n v_add_f32_e64 %6, %6, %8\n v_add_f32_e64 %7, %7, %8\n" : ACC : "v"(d)))
PROBE(k_add_pair, 8, asm volatile("v_add_f32 %0, %0, %1\n v_add_f32 %1, %1, %2\n v_add_f32 %2, %2, %3\n v_add_f32 %3, %3, %4\n"
                                  "v_add_f32 %4, %4, %5\n v_add_f32 %5, %5, %6\n v_add_f32 %6, %6, %7\n v_add_f32 %7, %7, %8\n" : ACC : "v"(d)))
PROBE(k_mul, 8, asm volatile("v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n"
                             "v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8\n" : ACC : "v"(c)))
PROBE(k_max, 8, asm volatile("v_max_f32 %0, %0, %8\n v_max_f32 %1, %1, %8\n v_max_f32 %2, %2, %8\n v_max_f32 %3, %3, %8\n"
                             "v_max_f32 %4, %4, %8\n v_max_f32 %5, %5, %8\n v_max_f32 %6, %6, %8\n v_max_f32 %7, %7, %8\n" : ACC : "v"(d)))
PROBE(k_add_dpp, 8, asm volatile("v_add_f32_dpp %0, %0, %0" DPP8 "\n v_add_f32_dpp %1, %1, %1" DPP8 "\n v_add_f32_dpp %2, %2, %2" DPP8 "\n v_add_f32_dpp %3, %3, %3" DPP8 "\n"
                                 "v_add_f32_dpp %4, %4, %4" DPP8 "\n v_add_f32_dpp %5, %5, %5" DPP8 "\n v_add_f32_dpp %6, %6, %6" DPP8 "\n v_add_f32_dpp %7, %7, %7" DPP8 "\n" : ACC))
PROBE(k_add_dpp_bm, 8, asm volatile("v_add_f32_dpp %0, %1, %1" DPPC "\n v_add_f32_dpp %1, %2, %2" DPPC "\n v_add_f32_dpp %2, %3, %3" DPPC "\n v_add_f32_dpp %3, %4, %4" DPPC "\n"
                                    "v_add_f32_dpp %4, %5, %5" DPPC "\n v_add_f32_dpp %5, %6, %6" DPPC "\n v_add_f32_dpp %6, %7, %7" DPPC "\n v_add_f32_dpp %7, %0, %0" DPPC "\n" : ACC))
PROBE(k_add_dpp_qp, 8, asm volatile("v_add_f32_dpp %0, %0, %0" QP "\n v_add_f32_dpp %1, %1, %1" QP "\n v_add_f32_dpp %2, %2, %2" QP "\n v_add_f32_dpp %3, %3, %3" QP "\n"
                                    "v_add_f32_dpp %4, %4, %4" QP "\n v_add_f32_dpp %5, %5, %5" QP "\n v_add_f32_dpp %6, %6, %6" QP "\n v_add_f32_dpp %7, %7, %7" QP "\n" : ACC))
PROBE(k_add_dpp_shr, 8, asm volatile("v_add_f32_dpp %0, %0, %0" SHR1 "\n v_add_f32_dpp %1, %1, %1" SHR1 "\n v_add_f32_dpp %2, %2, %2" SHR1 "\n v_add_f32_dpp %3, %3, %3" SHR1 "\n"
                                     "v_add_f32_dpp %4, %4, %4" SHR1 "\n v_add_f32_dpp %5, %5, %5" SHR1 "\n v_add_f32_dpp %6, %6, %6" SHR1 "\n v_add_f32_dpp %7, %7, %7" SHR1 "\n" : ACC))
// dst += dpp(src0) * src1 with src1 = 1.0: the same sum, the same rounding
PROBE(k_fmac_dpp, 8, asm volatile("v_fmac_f32_dpp %0, %0, %8" DPP8 "\n v_fmac_f32_dpp %1, %1, %8" DPP8 "\n v_fmac_f32_dpp %2, %2, %8" DPP8 "\n v_fmac_f32_dpp %3, %3, %8" DPP8 "\n"
                                  "v_fmac_f32_dpp %4, %4, %8" DPP8 "\n v_fmac_f32_dpp %5, %5, %8" DPP8 "\n v_fmac_f32_dpp %6, %6, %8" DPP8 "\n v_fmac_f32_dpp %7, %7, %8" DPP8 "\n" : ACC : "v"(c)))
PROBE(k_mov_dpp, 8, asm volatile("v_mov_b32_dpp %0, %1" DPP8 "\n v_mov_b32_dpp %1, %2" DPP8 "\n v_mov_b32_dpp %2, %3" DPP8 "\n v_mov_b32_dpp %3, %4" DPP8 "\n"
                                 "v_mov_b32_dpp %4, %5" DPP8 "\n v_mov_b32_dpp %5, %6" DPP8 "\n v_mov_b32_dpp %6, %7" DPP8 "\n v_mov_b32_dpp %7, %0" DPP8 "\n" : ACC))
PROBE(k_mov, 8, asm volatile("v_mov_b32 %0, %1\n v_mov_b32 %1, %2\n v_mov_b32 %2, %3\n v_mov_b32 %3, %4\n"
                             "v_mov_b32 %4, %5\n v_mov_b32 %5, %6\n v_mov_b32 %6, %7\n v_mov_b32 %7, %0\n" : ACC))
PROBE(k_cndmask, 8, asm volatile("v_cmp_lt_f32 vcc, %0, %8\n"
                                 "v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n v_cndmask_b32 %4, %4, %8, vcc\n"
                                 "v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc\n" : ACC : "v"(d) : "vcc"))
PROBE(k_cmp, 8, asm volatile("v_cmp_lt_f32 vcc, %0, %8\n v_cmp_lt_f32 vcc, %1, %8\n v_cmp_lt_f32 vcc, %2, %8\n v_cmp_lt_f32 vcc, %3, %8\n"
                             "v_cmp_lt_f32 vcc, %4, %8\n v_cmp_lt_f32 vcc, %5, %8\n v_cmp_lt_f32 vcc, %6, %8\n v_cmp_lt_f32 vcc, %7, %8\n" : ACC : "v"(d) : "vcc"))
PROBE(k_rcp, 8, asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n"
                             "v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n v_rcp_f32 %6, %6\n v_rcp_f32 %7, %7\n" : ACC))
PROBE(k_exp, 8, asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n"
                             "v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7\n" : ACC))
PROBE(k_and, 8, asm volatile("v_and_b32 %0, %0, %8\n v_and_b32 %1, %1, %8\n v_and_b32 %2, %2, %8\n v_and_b32 %3, %3, %8\n"
                             "v_and_b32 %4, %4, %8\n v_and_b32 %5, %5, %8\n v_and_b32 %6, %6, %8\n v_and_b32 %7, %7, %8\n" : ACC : "v"(d)))
PROBE(k_addu, 8, asm volatile("v_add_u32 %0, %0, %8\n v_add_u32 %1, %1, %8\n v_add_u32 %2, %2, %8\n v_add_u32 %3, %3, %8\n"
                              "v_add_u32 %4, %4, %8\n v_add_u32 %5, %5, %8\n v_add_u32 %6, %6, %8\n v_add_u32 %7, %7, %8\n" : ACC : "v"(d)))
// is a transcendental issued beside the fp32 pipe?  2 v_rcp + 6 v_fma per trip
PROBE(k_fma6_rcp2, 8, asm volatile("v_rcp_f32 %0, %0\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                                   "v_rcp_f32 %4, %4\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n" : ACC : "v"(c), "v"(d)))
// 4 fma + 4 dpp adds interleaved (do the two overlap?)
PROBE(k_fma4_dpp4, 8, asm volatile("v_fma_f32 %0, %0, %8, %9\n v_add_f32_dpp %1, %1, %1" DPP8 "\n v_fma_f32 %2, %2, %8, %9\n v_add_f32_dpp %3, %3, %3" DPP8 "\n"
                                   "v_fma_f32 %4, %4, %8, %9\n v_add_f32_dpp %5, %5, %5" DPP8 "\n v_fma_f32 %6, %6, %8, %9\n v_add_f32_dpp %7, %7, %7" DPP8 "\n" : ACC : "v"(c), "v"(d)))
// scalar work between vector instructions: does the SALU issue beside the VALU of the same wave / other waves?
PROBE(k_fma8_salu8, 8, asm volatile("v_fma_f32 %0, %0, %8, %9\n s_add_u32 s20, s20, 1\n v_fma_f32 %1, %1, %8, %9\n s_add_u32 s21, s21, 1\n v_fma_f32 %2, %2, %8, %9\n s_add_u32 s22, s22, 1\n"
                                    "v_fma_f32 %3, %3, %8, %9\n s_add_u32 s23, s23, 1\n v_fma_f32 %4, %4, %8, %9\n s_add_u32 s20, s20, 1\n v_fma_f32 %5, %5, %8, %9\n s_add_u32 s21, s21, 1\n"
                                    "v_fma_f32 %6, %6, %8, %9\n s_add_u32 s22, s22, 1\n v_fma_f32 %7, %7, %8, %9\n s_add_u32 s23, s23, 1\n" : ACC : "v"(c), "v"(d) : "s20", "s21", "s22", "s23", "scc"))

PROBE(k_add_sgpr, 8, asm volatile("v_add_f32 %0, s20, %0\n v_add_f32 %1, s20, %1\n v_add_f32 %2, s20, %2\n v_add_f32 %3, s20, %3\n v_add_f32 %4, s20, %4\n v_add_f32 %5, s20, %5\n v_add_f32 %6, s20, %6\n v_add_f32 %7, s20, %7\n " : ACC : "v"(c), "v"(d) : "s20"))
PROBE(k_fma_sgpr, 8, asm volatile("v_fma_f32 %0, %0, s20, %9\n v_fma_f32 %1, %1, s20, %9\n v_fma_f32 %2, %2, s20, %9\n v_fma_f32 %3, %3, s20, %9\n v_fma_f32 %4, %4, s20, %9\n v_fma_f32 %5, %5, s20, %9\n v_fma_f32 %6, %6, s20, %9\n v_fma_f32 %7, %7, s20, %9\n " : ACC : "v"(c), "v"(d) : "s20"))
PROBE(k_mov_sgpr, 8, asm volatile("v_mov_b32 %0, s20\n v_mov_b32 %1, s20\n v_mov_b32 %2, s20\n v_mov_b32 %3, s20\n v_mov_b32 %4, s20\n v_mov_b32 %5, s20\n v_mov_b32 %6, s20\n v_mov_b32 %7, s20\n " : ACC : "v"(c), "v"(d) : "s20"))
PROBE(k_cnd31, 33, asm volatile("v_cmp_lt_f32 vcc, %0, %9\n v_cndmask_b32 %0, %0, %9, vcc\n v_cndmask_b32 %1, %1, %9, vcc\n v_cndmask_b32 %2, %2, %9, vcc\n v_cndmask_b32 %3, %3, %9, vcc\n v_cndmask_b32 %4, %4, %9, vcc\n v_cndmask_b32 %5, %5, %9, vcc\n v_cndmask_b32 %6, %6, %9, vcc\n v_cndmask_b32 %7, %7, %9, vcc\n v_cndmask_b32 %0, %0, %9, vcc\n v_cndmask_b32 %1, %1, %9, vcc\n v_cndmask_b32 %2, %2, %9, vcc\n v_cndmask_b32 %3, %3, %9, vcc\n v_cndmask_b32 %4, %4, %9, vcc\n v_cndmask_b32 %5, %5, %9, vcc\n v_cndmask_b32 %6, %6, %9, vcc\n v_cndmask_b32 %7, %7, %9, vcc\n v_cndmask_b32 %0, %0, %9, vcc\n v_cndmask_b32 %1, %1, %9, vcc\n v_cndmask_b32 %2, %2, %9, vcc\n v_cndmask_b32 %3, %3, %9, vcc\n v_cndmask_b32 %4, %4, %9, vcc\n v_cndmask_b32 %5, %5, %9, vcc\n v_cndmask_b32 %6, %6, %9, vcc\n v_cndmask_b32 %7, %7, %9, vcc\n v_cndmask_b32 %0, %0, %9, vcc\n v_cndmask_b32 %1, %1, %9, vcc\n v_cndmask_b32 %2, %2, %9, vcc\n v_cndmask_b32 %3, %3, %9, vcc\n v_cndmask_b32 %4, %4, %9, vcc\n v_cndmask_b32 %5, %5, %9, vcc\n v_cndmask_b32 %6, %6, %9, vcc\n v_cndmask_b32 %7, %7, %9, vcc\n " : ACC : "v"(c), "v"(d) : "vcc"))
PROBE(k_cnd_e64, 33, asm volatile("v_cmp_lt_f32 s[20:21], %0, %9\n" "v_cndmask_b32_e64 %0, %0, %9, s[20:21]\n v_cndmask_b32_e64 %1, %1, %9, s[20:21]\n v_cndmask_b32_e64 %2, %2, %9, s[20:21]\n v_cndmask_b32_e64 %3, %3, %9, s[20:21]\n v_cndmask_b32_e64 %4, %4, %9, s[20:21]\n v_cndmask_b32_e64 %5, %5, %9, s[20:21]\n v_cndmask_b32_e64 %6, %6, %9, s[20:21]\n v_cndmask_b32_e64 %7, %7, %9, s[20:21]\n " "v_cndmask_b32_e64 %0, %0, %9, s[20:21]\n v_cndmask_b32_e64 %1, %1, %9, s[20:21]\n v_cndmask_b32_e64 %2, %2, %9, s[20:21]\n v_cndmask_b32_e64 %3, %3, %9, s[20:21]\n v_cndmask_b32_e64 %4, %4, %9, s[20:21]\n v_cndmask_b32_e64 %5, %5, %9, s[20:21]\n v_cndmask_b32_e64 %6, %6, %9, s[20:21]\n v_cndmask_b32_e64 %7, %7, %9, s[20:21]\n " "v_cndmask_b32_e64 %0, %0, %9, s[20:21]\n v_cndmask_b32_e64 %1, %1, %9, s[20:21]\n v_cndmask_b32_e64 %2, %2, %9, s[20:21]\n v_cndmask_b32_e64 %3, %3, %9, s[20:21]\n v_cndmask_b32_e64 %4, %4, %9, s[20:21]\n v_cndmask_b32_e64 %5, %5, %9, s[20:21]\n v_cndmask_b32_e64 %6, %6, %9, s[20:21]\n v_cndmask_b32_e64 %7, %7, %9, s[20:21]\n " "v_cndmask_b32_e64 %0, %0, %9, s[20:21]\n v_cndmask_b32_e64 %1, %1, %9, s[20:21]\n v_cndmask_b32_e64 %2, %2, %9, s[20:21]\n v_cndmask_b32_e64 %3, %3, %9, s[20:21]\n v_cndmask_b32_e64 %4, %4, %9, s[20:21]\n v_cndmask_b32_e64 %5, %5, %9, s[20:21]\n v_cndmask_b32_e64 %6, %6, %9, s[20:21]\n v_cndmask_b32_e64 %7, %7, %9, s[20:21]\n " : ACC : "v"(c), "v"(d) : "s20", "s21"))
PROBE(k_cmp_cnd_alt, 16, asm volatile("v_cmp_lt_f32 vcc, %0, %9\n v_cndmask_b32 %4, %4, %9, vcc\n v_cmp_lt_f32 vcc, %1, %9\n v_cndmask_b32 %5, %5, %9, vcc\n v_cmp_lt_f32 vcc, %2, %9\n v_cndmask_b32 %6, %6, %9, vcc\n v_cmp_lt_f32 vcc, %3, %9\n v_cndmask_b32 %7, %7, %9, vcc\n v_cmp_lt_f32 vcc, %4, %9\n v_cndmask_b32 %0, %0, %9, vcc\n v_cmp_lt_f32 vcc, %5, %9\n v_cndmask_b32 %1, %1, %9, vcc\n v_cmp_lt_f32 vcc, %6, %9\n v_cndmask_b32 %2, %2, %9, vcc\n v_cmp_lt_f32 vcc, %7, %9\n v_cndmask_b32 %3, %3, %9, vcc\n " : ACC : "v"(c), "v"(d) : "vcc"))
PROBE(k_cmp_e64, 8, asm volatile("v_cmp_lt_f32_e64 s[20:21], %0, %9\n v_cmp_lt_f32_e64 s[20:21], %1, %9\n v_cmp_lt_f32_e64 s[20:21], %2, %9\n v_cmp_lt_f32_e64 s[20:21], %3, %9\n v_cmp_lt_f32_e64 s[20:21], %4, %9\n v_cmp_lt_f32_e64 s[20:21], %5, %9\n v_cmp_lt_f32_e64 s[20:21], %6, %9\n v_cmp_lt_f32_e64 s[20:21], %7, %9\n " : ACC : "v"(c), "v"(d) : "s20", "s21"))
PROBE(k_min, 8, asm volatile("v_min_f32 %0, %0, %9\n v_min_f32 %1, %1, %9\n v_min_f32 %2, %2, %9\n v_min_f32 %3, %3, %9\n v_min_f32 %4, %4, %9\n v_min_f32 %5, %5, %9\n v_min_f32 %6, %6, %9\n v_min_f32 %7, %7, %9\n " : ACC : "v"(c), "v"(d)))
PROBE(k_max3, 8, asm volatile("v_max3_f32 %0, %0, %8, %9\n v_max3_f32 %1, %1, %8, %9\n v_max3_f32 %2, %2, %8, %9\n v_max3_f32 %3, %3, %8, %9\n v_max3_f32 %4, %4, %8, %9\n v_max3_f32 %5, %5, %8, %9\n v_max3_f32 %6, %6, %8, %9\n v_max3_f32 %7, %7, %8, %9\n " : ACC : "v"(c), "v"(d)))
PROBE(k_med3, 8, asm volatile("v_med3_f32 %0, %0, %8, %9\n v_med3_f32 %1, %1, %8, %9\n v_med3_f32 %2, %2, %8, %9\n v_med3_f32 %3, %3, %8, %9\n v_med3_f32 %4, %4, %8, %9\n v_med3_f32 %5, %5, %8, %9\n v_med3_f32 %6, %6, %8, %9\n v_med3_f32 %7, %7, %8, %9\n " : ACC : "v"(c), "v"(d)))
PROBE(k_mul_e64neg, 8, asm volatile("v_mul_f32_e64 %0, -%0, %8\n v_mul_f32_e64 %1, -%1, %8\n v_mul_f32_e64 %2, -%2, %8\n v_mul_f32_e64 %3, -%3, %8\n v_mul_f32_e64 %4, -%4, %8\n v_mul_f32_e64 %5, -%5, %8\n v_mul_f32_e64 %6, -%6, %8\n v_mul_f32_e64 %7, -%7, %8\n " : ACC : "v"(c), "v"(d)))
PROBE(k_fma_neg, 8, asm volatile("v_fma_f32 %0, -%0, %8, -%9\n v_fma_f32 %1, -%1, %8, -%9\n v_fma_f32 %2, -%2, %8, -%9\n v_fma_f32 %3, -%3, %8, -%9\n v_fma_f32 %4, -%4, %8, -%9\n v_fma_f32 %5, -%5, %8, -%9\n v_fma_f32 %6, -%6, %8, -%9\n v_fma_f32 %7, -%7, %8, -%9\n " : ACC : "v"(c), "v"(d)))
PROBE(k_bfe, 8, asm volatile("v_bfe_u32 %0, %0, 3, 5\n v_bfe_u32 %1, %1, 3, 5\n v_bfe_u32 %2, %2, 3, 5\n v_bfe_u32 %3, %3, 3, 5\n v_bfe_u32 %4, %4, 3, 5\n v_bfe_u32 %5, %5, 3, 5\n v_bfe_u32 %6, %6, 3, 5\n v_bfe_u32 %7, %7, 3, 5\n " : ACC : "v"(c), "v"(d)))
PROBE(k_bcnt, 8, asm volatile("v_bcnt_u32_b32 %0, %0, %9\n v_bcnt_u32_b32 %1, %1, %9\n v_bcnt_u32_b32 %2, %2, %9\n v_bcnt_u32_b32 %3, %3, %9\n v_bcnt_u32_b32 %4, %4, %9\n v_bcnt_u32_b32 %5, %5, %9\n v_bcnt_u32_b32 %6, %6, %9\n v_bcnt_u32_b32 %7, %7, %9\n " : ACC : "v"(c), "v"(d)))
PROBE(k_lshl, 8, asm volatile("v_lshlrev_b32 %0, 1, %0\n v_lshlrev_b32 %1, 1, %1\n v_lshlrev_b32 %2, 1, %2\n v_lshlrev_b32 %3, 1, %3\n v_lshlrev_b32 %4, 1, %4\n v_lshlrev_b32 %5, 1, %5\n v_lshlrev_b32 %6, 1, %6\n v_lshlrev_b32 %7, 1, %7\n " : ACC : "v"(c), "v"(d)))
PROBE(k_add3, 8, asm volatile("v_add3_u32 %0, %0, %8, %9\n v_add3_u32 %1, %1, %8, %9\n v_add3_u32 %2, %2, %8, %9\n v_add3_u32 %3, %3, %8, %9\n v_add3_u32 %4, %4, %8, %9\n v_add3_u32 %5, %5, %8, %9\n v_add3_u32 %6, %6, %8, %9\n v_add3_u32 %7, %7, %8, %9\n " : ACC : "v"(c), "v"(d)))
PROBE(k_mad24, 8, asm volatile("v_mad_u32_u24 %0, %0, %8, %9\n v_mad_u32_u24 %1, %1, %8, %9\n v_mad_u32_u24 %2, %2, %8, %9\n v_mad_u32_u24 %3, %3, %8, %9\n v_mad_u32_u24 %4, %4, %8, %9\n v_mad_u32_u24 %5, %5, %8, %9\n v_mad_u32_u24 %6, %6, %8, %9\n v_mad_u32_u24 %7, %7, %8, %9\n " : ACC : "v"(c), "v"(d)))
PROBE(k_mullo, 8, asm volatile("v_mul_lo_u32 %0, %0, %8\n v_mul_lo_u32 %1, %1, %8\n v_mul_lo_u32 %2, %2, %8\n v_mul_lo_u32 %3, %3, %8\n v_mul_lo_u32 %4, %4, %8\n v_mul_lo_u32 %5, %5, %8\n v_mul_lo_u32 %6, %6, %8\n v_mul_lo_u32 %7, %7, %8\n " : ACC : "v"(c), "v"(d)))
PROBE(k_or3, 8, asm volatile("v_or3_b32 %0, %0, %8, %9\n v_or3_b32 %1, %1, %8, %9\n v_or3_b32 %2, %2, %8, %9\n v_or3_b32 %3, %3, %8, %9\n v_or3_b32 %4, %4, %8, %9\n v_or3_b32 %5, %5, %8, %9\n v_or3_b32 %6, %6, %8, %9\n v_or3_b32 %7, %7, %8, %9\n " : ACC : "v"(c), "v"(d)))
PROBE(k_cvt, 8, asm volatile("v_cvt_f32_i32 %0, %0\n v_cvt_f32_i32 %1, %1\n v_cvt_f32_i32 %2, %2\n v_cvt_f32_i32 %3, %3\n v_cvt_f32_i32 %4, %4\n v_cvt_f32_i32 %5, %5\n v_cvt_f32_i32 %6, %6\n v_cvt_f32_i32 %7, %7\n " : ACC : "v"(c), "v"(d)))
PROBE(k_rdlane, 8, asm volatile("v_readlane_b32 s20, %0, 3\n v_readlane_b32 s20, %1, 3\n v_readlane_b32 s20, %2, 3\n v_readlane_b32 s20, %3, 3\n v_readlane_b32 s20, %4, 3\n v_readlane_b32 s20, %5, 3\n v_readlane_b32 s20, %6, 3\n v_readlane_b32 s20, %7, 3\n " : ACC : "v"(c), "v"(d) : "s20"))
PROBE(k_rdfirst, 8, asm volatile("v_readfirstlane_b32 s20, %0\n v_readfirstlane_b32 s20, %1\n v_readfirstlane_b32 s20, %2\n v_readfirstlane_b32 s20, %3\n v_readfirstlane_b32 s20, %4\n v_readfirstlane_b32 s20, %5\n v_readfirstlane_b32 s20, %6\n v_readfirstlane_b32 s20, %7\n " : ACC : "v"(c), "v"(d) : "s20"))
PROBE(k_sdwa, 8, asm volatile("v_add_u32_sdwa %0, %0, %9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:DWORD\n v_add_u32_sdwa %1, %1, %9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:DWORD\n v_add_u32_sdwa %2, %2, %9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:DWORD\n v_add_u32_sdwa %3, %3, %9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:DWORD\n v_add_u32_sdwa %4, %4, %9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:DWORD\n v_add_u32_sdwa %5, %5, %9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:DWORD\n v_add_u32_sdwa %6, %6, %9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:DWORD\n v_add_u32_sdwa %7, %7, %9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:DWORD\n " : ACC : "v"(c), "v"(d)))
PROBE(k_salu8, 8, asm volatile("s_add_u32 s20, s20, 1\n s_add_u32 s21, s21, 1\n s_add_u32 s22, s22, 1\n s_add_u32 s23, s23, 1\n s_add_u32 s24, s24, 1\n s_add_u32 s25, s25, 1\n s_add_u32 s26, s26, 1\n s_add_u32 s27, s27, 1\n " : ACC : "v"(c), "v"(d) : "s20","s21","s22","s23","s24","s25","s26","s27","scc"))
PROBE(k_saveexec8, 16, asm volatile("s_and_saveexec_b64 s[20:21], exec\n s_or_b64 exec, exec, s[20:21]\n s_and_saveexec_b64 s[20:21], exec\n s_or_b64 exec, exec, s[20:21]\n s_and_saveexec_b64 s[20:21], exec\n s_or_b64 exec, exec, s[20:21]\n s_and_saveexec_b64 s[20:21], exec\n s_or_b64 exec, exec, s[20:21]\n s_and_saveexec_b64 s[20:21], exec\n s_or_b64 exec, exec, s[20:21]\n s_and_saveexec_b64 s[20:21], exec\n s_or_b64 exec, exec, s[20:21]\n s_and_saveexec_b64 s[20:21], exec\n s_or_b64 exec, exec, s[20:21]\n s_and_saveexec_b64 s[20:21], exec\n s_or_b64 exec, exec, s[20:21]\n " : ACC : "v"(c), "v"(d) : "s20","s21","scc"))
PROBE(k_fma8_wait8, 8, asm volatile("v_fma_f32 %0, %0, %8, %9\n s_waitcnt lgkmcnt(0)\n v_fma_f32 %1, %1, %8, %9\n s_waitcnt lgkmcnt(0)\n v_fma_f32 %2, %2, %8, %9\n s_waitcnt lgkmcnt(0)\n v_fma_f32 %3, %3, %8, %9\n s_waitcnt lgkmcnt(0)\n v_fma_f32 %4, %4, %8, %9\n s_waitcnt lgkmcnt(0)\n v_fma_f32 %5, %5, %8, %9\n s_waitcnt lgkmcnt(0)\n v_fma_f32 %6, %6, %8, %9\n s_waitcnt lgkmcnt(0)\n v_fma_f32 %7, %7, %8, %9\n s_waitcnt lgkmcnt(0)\n " : ACC : "v"(c), "v"(d)))
PROBE(k_fma8_nop8, 8, asm volatile("v_fma_f32 %0, %0, %8, %9\n s_nop 1\n v_fma_f32 %1, %1, %8, %9\n s_nop 1\n v_fma_f32 %2, %2, %8, %9\n s_nop 1\n v_fma_f32 %3, %3, %8, %9\n s_nop 1\n v_fma_f32 %4, %4, %8, %9\n s_nop 1\n v_fma_f32 %5, %5, %8, %9\n s_nop 1\n v_fma_f32 %6, %6, %8, %9\n s_nop 1\n v_fma_f32 %7, %7, %8, %9\n s_nop 1\n " : ACC : "v"(c), "v"(d)))
PROBE(k_fma8_salu2, 8, asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n s_add_u32 s20, s20, 1\n v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n s_add_u32 s21, s21, 1\n" : ACC : "v"(c), "v"(d) : "s20","s21","scc"))
PROBE(k_fma8_salu4, 8, asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n s_add_u32 s20, s20, 1\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n s_add_u32 s21, s21, 1\n v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n s_add_u32 s22, s22, 1\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n s_add_u32 s23, s23, 1\n" : ACC : "v"(c), "v"(d) : "s20","s21","s22","s23","scc"))
PROBE(k_and_mask, 8, asm volatile("v_and_b32 %0, %9, %0\n v_and_b32 %1, %9, %1\n v_and_b32 %2, %9, %2\n v_and_b32 %3, %9, %3\n v_and_b32 %4, %9, %4\n v_and_b32 %5, %9, %5\n v_and_b32 %6, %9, %6\n v_and_b32 %7, %9, %7\n " : ACC : "v"(c), "v"(d)))
// one dependent chain (run with W = 1 for the latency of a dependent instruction; the 8 accumulators are chained a0 -> a1 -> ... -> a0)
PROBE(k_fma_dep, 8, asm volatile("v_fma_f32 %0, %7, %8, %9\n v_fma_f32 %1, %0, %8, %9\n v_fma_f32 %2, %1, %8, %9\n v_fma_f32 %3, %2, %8, %9\n"
                                 "v_fma_f32 %4, %3, %8, %9\n v_fma_f32 %5, %4, %8, %9\n v_fma_f32 %6, %5, %8, %9\n v_fma_f32 %7, %6, %8, %9\n" : ACC : "v"(c), "v"(d)))
PROBE(k_dpp_dep, 8, asm volatile("s_nop 1\n v_add_f32_dpp %0, %7, %7" DPP8 "\n s_nop 1\n v_add_f32_dpp %1, %0, %0" DPP8 "\n s_nop 1\n v_add_f32_dpp %2, %1, %1" DPP8 "\n s_nop 1\n v_add_f32_dpp %3, %2, %2" DPP8 "\n"
                                 "s_nop 1\n v_add_f32_dpp %4, %3, %3" DPP8 "\n s_nop 1\n v_add_f32_dpp %5, %4, %4" DPP8 "\n s_nop 1\n v_add_f32_dpp %6, %5, %5" DPP8 "\n s_nop 1\n v_add_f32_dpp %7, %6, %6" DPP8 "\n" : ACC))
PROBE(k_rcp_dep, 8, asm volatile("v_rcp_f32 %0, %7\n v_rcp_f32 %1, %0\n v_rcp_f32 %2, %1\n v_rcp_f32 %3, %2\n v_rcp_f32 %4, %3\n v_rcp_f32 %5, %4\n v_rcp_f32 %6, %5\n v_rcp_f32 %7, %6\n" : ACC))
PROBE(k_fma_dep2, 8, asm volatile("v_fma_f32 %0, %6, %8, %9\n v_fma_f32 %1, %7, %8, %9\n v_fma_f32 %2, %0, %8, %9\n v_fma_f32 %3, %1, %8, %9\n"
                                  "v_fma_f32 %4, %2, %8, %9\n v_fma_f32 %5, %3, %8, %9\n v_fma_f32 %6, %4, %8, %9\n v_fma_f32 %7, %5, %8, %9\n" : ACC : "v"(c), "v"(d)))
// LDS: 8 x ds_read_b128 per trip, 4 distinct addresses per wave (one per DPP row: the walks' broadcast reads of a staged record)
typedef float f4 __attribute__((ext_vector_type(4)));
PROBE(k_ds128_row, 8, { f4 t0, t1, t2, t3, t4, t5, t6, t7; const unsigned ad = (threadIdx.x >> 4) * 80u;
                        asm volatile("ds_read_b128 %0, %8\n ds_read_b128 %1, %8 offset:16\n ds_read_b128 %2, %8 offset:32\n ds_read_b128 %3, %8 offset:48\n"
                                     "ds_read_b128 %4, %8 offset:64\n ds_read_b128 %5, %8 offset:5120\n ds_read_b128 %6, %8 offset:5136\n ds_read_b128 %7, %8 offset:5152\n s_waitcnt lgkmcnt(0)\n"
                                     : "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3), "=&v"(t4), "=&v"(t5), "=&v"(t6), "=&v"(t7) : "v"(ad) : "memory");
                        a0 += t0.x; a1 += t1.y; a2 += t2.z; a3 += t3.w; a4 += t4.x; a5 += t5.y; a6 += t6.z; a7 += t7.w; })
PROBE(k_ds128_lane, 8, { f4 t0, t1, t2, t3, t4, t5, t6, t7; const unsigned ad = threadIdx.x * 16u;
                         asm volatile("ds_read_b128 %0, %8\n ds_read_b128 %1, %8 offset:16384\n ds_read_b128 %2, %8 offset:32768\n ds_read_b128 %3, %8 offset:49152\n"
                                      "ds_read_b128 %4, %8 offset:64\n ds_read_b128 %5, %8 offset:16448\n ds_read_b128 %6, %8 offset:32832\n ds_read_b128 %7, %8 offset:49216\n s_waitcnt lgkmcnt(0)\n"
                                      : "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3), "=&v"(t4), "=&v"(t5), "=&v"(t6), "=&v"(t7) : "v"(ad) : "memory");
                         a0 += t0.x; a1 += t1.y; a2 += t2.z; a3 += t3.w; a4 += t4.x; a5 += t5.y; a6 += t6.z; a7 += t7.w; })
PROBE(k_ds32_row, 8, { float t0, t1, t2, t3, t4, t5, t6, t7; const unsigned ad = (threadIdx.x >> 4) * 80u;
                        asm volatile("ds_read_b32 %0, %8\n ds_read_b32 %1, %8 offset:16\n ds_read_b32 %2, %8 offset:32\n ds_read_b32 %3, %8 offset:48\n"
                                     "ds_read_b32 %4, %8 offset:64\n ds_read_b32 %5, %8 offset:5120\n ds_read_b32 %6, %8 offset:5136\n ds_read_b32 %7, %8 offset:5152\n s_waitcnt lgkmcnt(0)\n"
                                     : "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3), "=&v"(t4), "=&v"(t5), "=&v"(t6), "=&v"(t7) : "v"(ad) : "memory");
                        a0 += t0; a1 += t1; a2 += t2; a3 += t3; a4 += t4; a5 += t5; a6 += t6; a7 += t7; })
// the compiler's usual select: v_cmp -> a few unrelated instructions -> v_cndmask reading vcc (e32) or an SGPR pair (e64)
PROBE(k_cmp3cnd2_vcc, 8, asm volatile("v_cmp_lt_f32 vcc, %0, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                                      "v_cndmask_b32 %4, %4, %9, vcc\n v_cndmask_b32 %5, %5, %9, vcc\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n" : ACC : "v"(c), "v"(d) : "vcc"))
PROBE(k_cmp3cnd2_sgpr, 8, asm volatile("v_cmp_lt_f32 s[20:21], %0, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                                       "v_cndmask_b32_e64 %4, %4, %9, s[20:21]\n v_cndmask_b32_e64 %5, %5, %9, s[20:21]\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n" : ACC : "v"(c), "v"(d) : "s20", "s21"))
PROBE(k_cmp3cnd2_vcc64, 8, asm volatile("v_cmp_lt_f32 vcc, %0, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                                        "v_cndmask_b32_e64 %4, %4, %9, vcc\n v_cndmask_b32_e64 %5, %5, %9, vcc\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n" : ACC : "v"(c), "v"(d) : "vcc"))
PROBE(k_cmp1cnd1_vcc, 8, asm volatile("v_cmp_lt_f32 vcc, %0, %9\n v_fma_f32 %1, %1, %8, %9\n v_cndmask_b32 %4, %4, %9, vcc\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                                      "v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n" : ACC : "v"(c), "v"(d) : "vcc"))
PROBE(k_cmp0cnd4_vcc, 8, asm volatile("v_cmp_lt_f32 vcc, %0, %9\n v_cndmask_b32 %1, %1, %9, vcc\n v_cndmask_b32 %2, %2, %9, vcc\n v_cndmask_b32 %3, %3, %9, vcc\n v_cndmask_b32 %4, %4, %9, vcc\n"
                                      "v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n" : ACC : "v"(c), "v"(d) : "vcc"))
PROBE(k_pk_fma, 4, { typedef float f2 __attribute__((ext_vector_type(2))); f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}; const f2 pc = {c, c}, pd = {d, d};
                     asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pc), "v"(pd));
                     a0 = p0.x; a1 = p0.y; a2 = p1.x; a3 = p1.y; a4 = p2.x; a5 = p2.y; a6 = p3.x; a7 = p3.y; })
PROBE(k_pk_mul, 4, { typedef float f2 __attribute__((ext_vector_type(2))); f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}; const f2 pc = {c, c};
                     asm volatile("v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4\n" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pc));
                     a0 = p0.x; a1 = p0.y; a2 = p1.x; a3 = p1.y; a4 = p2.x; a5 = p2.y; a6 = p3.x; a7 = p3.y; })
PROBE(k_pk_add, 4, { typedef float f2 __attribute__((ext_vector_type(2))); f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}; const f2 pd = {d, d};
                     asm volatile("v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4\n" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pd));
                     a0 = p0.x; a1 = p0.y; a2 = p1.x; a3 = p1.y; a4 = p2.x; a5 = p2.y; a6 = p3.x; a7 = p3.y; })
PROBE(k_or, 8, asm volatile("v_or_b32 %0, %0, %9\n v_or_b32 %1, %1, %9\n v_or_b32 %2, %2, %9\n v_or_b32 %3, %3, %9\n v_or_b32 %4, %4, %9\n v_or_b32 %5, %5, %9\n v_or_b32 %6, %6, %9\n v_or_b32 %7, %7, %9\n" : ACC : "v"(c), "v"(d)))
PROBE(k_subu, 8, asm volatile("v_sub_u32 %0, %0, %9\n v_sub_u32 %1, %1, %9\n v_sub_u32 %2, %2, %9\n v_sub_u32 %3, %3, %9\n v_sub_u32 %4, %4, %9\n v_sub_u32 %5, %5, %9\n v_sub_u32 %6, %6, %9\n v_sub_u32 %7, %7, %9\n" : ACC : "v"(c), "v"(d)))
PROBE(k_lshr, 8, asm volatile("v_lshrrev_b32 %0, 1, %0\n v_lshrrev_b32 %1, 1, %1\n v_lshrrev_b32 %2, 1, %2\n v_lshrrev_b32 %3, 1, %3\n v_lshrrev_b32 %4, 1, %4\n v_lshrrev_b32 %5, 1, %5\n v_lshrrev_b32 %6, 1, %6\n v_lshrrev_b32 %7, 1, %7\n" : ACC))
PROBE(k_mov64, 8, { typedef float f2 __attribute__((ext_vector_type(2))); f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7};
                    asm volatile("v_mov_b64 %0, %1\n v_mov_b64 %1, %2\n v_mov_b64 %2, %3\n v_mov_b64 %3, %0\n v_mov_b64 %0, %1\n v_mov_b64 %1, %2\n v_mov_b64 %2, %3\n v_mov_b64 %3, %0\n" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3));
                    a0 = p0.x; a1 = p0.y; a2 = p1.x; a3 = p1.y; a4 = p2.x; a5 = p2.y; a6 = p3.x; a7 = p3.y; })

struct Kind { const char* name; void (*fn)(unsigned long long*, int, float, float); int ninst; float c, d; };

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 1024;
    const int wps = argc > 2 ? atoi(argv[2]) : 4;      // waves per SIMD
    const int threads = 256 * wps, waves = 4 * wps;
    const size_t lds = 100 * 1024;
    unsigned long long* buf = nullptr;
    const size_t words = 4 * (size_t)kBlocks * kMaxWaves + 4;
    CHECK(hipMalloc(&buf, sizeof(unsigned long long) * words));
    static unsigned long long host[4 * kBlocks * kMaxWaves + 4];
#define K(NAME, C, D) {#NAME, NAME, NAME##_n, C, D}
    const Kind kinds[] = {K(k_fma, 1.0001f, 1e-6f), K(k_fma_one, 0.f, 1e-6f), K(k_fma_3src, 0.f, 1e-30f), K(k_add_inl, 0.f, 0.f), K(k_sub, 0.f, 1e-6f), K(k_fmac, 1.0001f, 1e-6f), K(k_add, 0.f, 1e-6f), K(k_add_rev, 0.f, 1e-6f),
                          K(k_add_e64, 0.f, 1e-6f), K(k_add_pair, 0.f, 1e-30f), K(k_mul, 1.0001f, 0.f), K(k_max, 0.f, 1e-6f), K(k_add_dpp, 0.f, 0.f),
                          K(k_add_dpp_bm, 0.f, 0.f), K(k_add_dpp_qp, 0.f, 0.f), K(k_add_dpp_shr, 0.f, 0.f), K(k_fmac_dpp, 1.0f, 0.f), K(k_mov_dpp, 0.f, 0.f),
                          K(k_mov, 0.f, 0.f), K(k_cndmask, 0.f, 1e-6f), K(k_cmp, 0.f, 1e-6f), K(k_rcp, 0.f, 0.f), K(k_exp, 0.f, 0.f), K(k_and, 0.f, 1e-6f),
                          K(k_addu, 0.f, 1e-6f), K(k_fma6_rcp2, 1.0001f, 1e-6f), K(k_fma4_dpp4, 1.0001f, 1e-6f), K(k_fma8_salu8, 1.0001f, 1e-6f),
                          K(k_add_sgpr, 1.0001f, 1e-6f), K(k_fma_sgpr, 1.0001f, 1e-6f), K(k_mov_sgpr, 1.0001f, 1e-6f), K(k_cnd31, 1.0001f, 1e-6f), K(k_cnd_e64, 1.0001f, 1e-6f), K(k_cmp_cnd_alt, 1.0001f, 1e-6f), K(k_cmp_e64, 1.0001f, 1e-6f), K(k_min, 1.0001f, 1e-6f), K(k_max3, 1.0001f, 1e-6f), K(k_med3, 1.0001f, 1e-6f), K(k_mul_e64neg, 1.0001f, 1e-6f), K(k_fma_neg, 1.0001f, 1e-6f), K(k_bfe, 1.0001f, 1e-6f), K(k_bcnt, 1.0001f, 1e-6f), K(k_lshl, 1.0001f, 1e-6f), K(k_add3, 1.0001f, 1e-6f), K(k_mad24, 1.0001f, 1e-6f), K(k_mullo, 1.0001f, 1e-6f), K(k_or3, 1.0001f, 1e-6f), K(k_cvt, 1.0001f, 1e-6f), K(k_rdlane, 1.0001f, 1e-6f), K(k_rdfirst, 1.0001f, 1e-6f), K(k_sdwa, 1.0001f, 1e-6f), K(k_salu8, 1.0001f, 1e-6f), K(k_saveexec8, 1.0001f, 1e-6f), K(k_fma8_wait8, 1.0001f, 1e-6f), K(k_fma8_nop8, 1.0001f, 1e-6f), K(k_fma8_salu2, 1.0001f, 1e-6f), K(k_fma8_salu4, 1.0001f, 1e-6f), K(k_and_mask, 1.0001f, 1e-6f), K(k_cmp3cnd2_vcc, 1.0001f, 1e-6f), K(k_cmp3cnd2_sgpr, 1.0001f, 1e-6f), K(k_cmp3cnd2_vcc64, 1.0001f, 1e-6f), K(k_cmp1cnd1_vcc, 1.0001f, 1e-6f), K(k_cmp0cnd4_vcc, 1.0001f, 1e-6f), K(k_pk_fma, 1.0001f, 1e-6f), K(k_pk_mul, 1.0001f, 0.f), K(k_pk_add, 0.f, 1e-6f), K(k_or, 0.f, 0.f), K(k_subu, 0.f, 0.f), K(k_lshr, 0.f, 0.f), K(k_mov64, 0.f, 0.f), K(k_fma_dep, 1.0001f, 1e-6f), K(k_fma_dep2, 1.0001f, 1e-6f), K(k_dpp_dep, 0.f, 0.f), K(k_rcp_dep, 0.f, 0.f), K(k_ds128_row, 0.f, 0.f), K(k_ds128_lane, 0.f, 0.f), K(k_ds32_row, 0.f, 0.f), K(k_fma, 1.0001f, 1e-6f)};
    printf("# %d waves per SIMD, %d loop trips of 4 copies\n%-16s %14s %9s %12s %10s\n", wps, iters, "kind", "cyc/inst/SIMD", "(max)", "G inst/s", "clock GHz");
    for (const Kind& k : kinds) CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k.fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    for (int rep = 0; rep < 2; rep++) {
        for (const Kind& k : kinds) {
            hipLaunchKernelGGL(k.fn, dim3(kBlocks), dim3(threads), lds, 0, buf, 32, k.c, k.d);
            hipLaunchKernelGGL(k.fn, dim3(kBlocks), dim3(threads), lds, 0, buf, iters, k.c, k.d);
            CHECK(hipDeviceSynchronize());
            CHECK(hipMemcpy(host, buf, sizeof(unsigned long long) * words, hipMemcpyDeviceToHost));
            double cyc = 0.0, ticks = 0.0, cmax = 0.0;
            unsigned long long f0 = ~0ull, l0 = 0ull;
            for (int b = 0; b < kBlocks; b++)
                for (int w = 0; w < waves; w++) {
                    const unsigned long long* h = host + 4 + 4 * (b * kMaxWaves + w);
                    cyc += (double)h[0]; ticks += (double)h[1];
                    if ((double)h[0] > cmax) cmax = (double)h[0];
                    if (h[2] < f0) f0 = h[2];
                    if (h[3] > l0) l0 = h[3];
                }
            const double n_inst = (double)iters * k.ninst;
            const double insts = (double)kBlocks * waves * n_inst;
            const double span_ns = (double)(l0 - f0) * 10.0;
            if (rep == 1)
                printf("%-16s %14.3f %9.3f %12.1f %10.3f\n", k.name, cyc / (kBlocks * waves) / n_inst / wps, cmax / n_inst / wps, insts / span_ns, cyc / ticks * 0.1);
        }
    }
    (void)hipFree(buf);
    return 0;
}
