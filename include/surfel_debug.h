/*
 * surfel_debug.h — diagnostics of libsurfel_hip.so: stage timings, counters, white-box layouts and hardware probes used by tests/,
 * bench.py and scripts/.  Not part of the drop-in boundary (include/surfel_hip.h); no reference counterpart.
 */
#ifndef SURFEL_DEBUG_H
#define SURFEL_DEBUG_H

#include "surfel_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Per-stage device timings (ms) of the last forward / backward made with debug != 0 on this thread; returns the stages written. */
int surfel_last_stage_ms(float* ms, int cap);
int surfel_last_stage_ids(int* ids, int cap);
/* debug >= 2: stages are bracketed with HIP events WITHOUT synchronising; this resolves every pending pair into sum_ms[stage] /
 * count[stage] (cap >= 11) and returns the number of stage ids.  At most 8192 pending pairs are kept. */
int surfel_collect_stage_ms(float* sum_ms, int* count, int cap);
const char* surfel_stage_name(int stage);

/* The library's stable LSD radix sort of (u32 key, u32 value) pairs on key bits [begin_bit, end_bit), in place (n < 2^30). */
int surfel_debug_sort_pairs(surfel_alloc_fn scratch_alloc, void* scratch_user, uint32_t* keys, uint32_t* vals, int64_t n,
                            int begin_bit, int end_bit, void* stream);

/* A device buffer of 8 uint64 (caller-zeroed) that every following blend-backward launch accumulates into — [0] lane slots issued,
 * [1] lanes that held a composited pair, [2] wave visits, [3] (sub-tile | quad, instance) visits, [4] of those with a composited pair,
 * [5] quad variant: 4x4 sub-tiles with a composited pair — or NULL to switch the instrumented kernels off. */
int surfel_debug_set_blend_stats(void* dev_u64x8);

/* How the last forward of this thread sized its binning buffers: 0 exact, 1 capacity, 2 capacity overflowed and redone, 4 lazy count. */
int surfel_debug_last_binning(void);
/* How often this thread's per-frame-size history (16 sizes) had to drop a size (its next frame takes the exact path: speed only). */
int surfel_debug_capacity_evictions(void);
/* Byte offsets inside the image buffer of a width x height frame: out[0] total, [1] final_T / M1 / M2, [2] contributor planes, [3] tile map. */
int surfel_debug_image_layout(int width, int height, int64_t* out);

/* What THIS GPU sustains (csrc/box_probe.hip): out[0] us per dependent launch, [1] G wave-inst/s of an FMA grid, [2] shader clock GHz,
 * [3] ms of that grid, [4] cycles per wave-instruction per SIMD, [5] G wave-inst/s over the grid's span, [6] / [7] the same for v_add_f32 /
 * v_pk_fma_f32, [8] M visits/s of a frozen blend-like instruction mix.  scratch >= 128 KiB.  The call synchronises. */
int surfel_debug_box_probe(void* scratch, int64_t scratch_bytes, float* out9, void* stream);
/* Dependent-load latency: one lane chases `hops` loads through `bytes` (>= 1 MiB) of `buf`. */
int surfel_debug_latency_probe(void* buf, int64_t bytes, int hops, float* ns_per_hop, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SURFEL_DEBUG_H */
