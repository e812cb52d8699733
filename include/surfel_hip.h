/*
 * surfel_hip.h — C ABI of libsurfel_hip.so: the MI355X (gfx950) differentiable surfel rasterizer
 * and the simple-knn initialisation kernel.
 *
 * This is the drop-in boundary for the hot path of hbb1/2d-gaussian-splatting.  The reference binds
 * the same functionality through two pybind11 modules that are absent (un-vendored submodules,
 * /root/reference/.gitmodules:1-6):
 *     diff_surfel_rasterization._C : rasterize_gaussians / rasterize_gaussians_backward / mark_visible
 *     simple_knn._C                : distCUDA2
 * Their call sites in the reference are cited per function below.  Plain pointers and sizes only —
 * no torch types; every pointer is a DEVICE pointer unless stated; `stream` is a hipStream_t passed
 * as void* (NULL = default stream).  The library never calls hipMalloc: scratch memory is obtained
 * through caller-supplied allocator callbacks (the reference's std::function<char*(size_t)> "resize"
 * functors), so PyTorch's caching allocator owns all memory.
 *
 * Error convention: functions return a value >= 0 on success and a negative SURFEL_E_* code on
 * failure; surfel_last_error() returns a thread-local message.  With debug == 1 every stage is
 * followed by a stream synchronise + hipGetLastError (the reference's `debug` flag,
 * /root/reference/gaussian_renderer/__init__.py:49); debug == 2 only records timing events.
 */
#ifndef SURFEL_HIP_H
#define SURFEL_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SURFEL_ABI_VERSION 1

#define SURFEL_E_INVALID (-1)   /* bad argument combination / null pointer */
#define SURFEL_E_ALLOC   (-2)   /* allocator callback returned NULL */
#define SURFEL_E_HIP     (-3)   /* a HIP call or kernel failed (message has the hipError string) */
#define SURFEL_E_LIMIT   (-4)   /* size exceeds an internal limit (e.g. > 2^32-1 tile instances) */
#define SURFEL_E_OVERFLOW (-5)  /* a lazily counted frame (SURFEL_OPT_LAZY_COUNT) held more instances than its capacity: render it again */

/* The `debug` argument of the rasterizer entry points: low byte = debug mode (0 off, 1 synchronise + check after every stage,
 * 2 / 3 record stage timing events), upper bits = PER-CALL option overrides, so callers (tests above all) need not flip the
 * process-wide defaults of surfel_set_option().  All overrides leave results bit-identical. */
#define SURFEL_OPT_NO_CULL        (1 << 8)             /* forward: "cull" = 0 for this call */
#define SURFEL_OPT_TILE_SORT(m)   ((((m) + 1) & 3) << 9)   /* forward: "tile_depth_sort" = m (0, 1, 2) for this call */
#define SURFEL_OPT_BWD_QUAD       (1 << 11)            /* backward: per-quad walk ("bwd_variant" = 1) for this call */
#define SURFEL_OPT_BWD_ROWS       (1 << 12)            /* backward: per-row walk ("bwd_variant" = 0) for this call */
#define SURFEL_OPT_PBWD_COOP      (1 << 13)            /* backward: wave-cooperative gather of the gradient records in the per-surfel kernel (default: by rule, R >= 6 P and R >= 2^25) */
#define SURFEL_OPT_PBWD_THREAD    (1 << 14)            /* backward: per-thread gather of the gradient records */
#define SURFEL_OPT_EXACT_BINNING  (1 << 16)            /* forward: "capacity_binning" = 0 for this call (binning buffers sized after a host wait for the instance count) */
#define SURFEL_OPT_TILE_CUTS      (1 << 17)            /* backward: no gradient records behind a tile's saturation point, preprocess_bwd tests the tile cuts (default: R >= 2^21) */
#define SURFEL_OPT_ZERO_RECORDS   (1 << 18)            /* backward: zero records behind a tile's saturation point (default: R < 2^21); bit-identical to the cuts */
#define SURFEL_OPT_TILE_ORDER(m)  ((((m) + 1) & 3) << 19)  /* forward: "tile_order" = m (0, 1, 2) for this call (the matching backward follows the forward) */
#define SURFEL_OPT_LAZY_COUNT     (1 << 21)            /* forward: do not wait for the instance count (see surfel_forward_count) */
#define SURFEL_OPT_BWD_GATHER     (1 << 22)            /* backward: ignore the forward's tile stream ("tile_stream") and gather the records by surfel id; bit-identical */
#define SURFEL_OPT_NO_STREAM      (1 << 23)            /* forward: no backward will follow this call (inference, no_grad renders): leave no tile stream behind (saves 84 B per instance of stores and memory); a backward handed such a frame gathers by surfel id, same bits */
#define SURFEL_OPT_BWD_SCAN       (1 << 15)            /* backward: scan walk ("bwd_variant" = 3) for this call; deterministic, NOT bit-identical to rows / quad */

/* Allocator callback: return a device pointer to `bytes` bytes, 256-byte aligned, valid until the
 * caller frees it.  Replaces the resize functors the reference's binding hands to the native
 * rasterizer (geometry / binning / image buffers; SURVEY.md §8b "ownership"). */
typedef void* (*surfel_alloc_fn)(void* user, size_t bytes);

int surfel_abi_version(void);
const char* surfel_last_error(void);

/*
 * Forward rasterisation.  Replaces `_C.rasterize_gaussians` (reference call site:
 * /root/reference/gaussian_renderer/__init__.py:97-106 through GaussianRasterizer.forward).
 *
 *   P surfels, D active SH degree (0..3), M SH coefficients stored per surfel (16).
 *   background[3]; means3D[P,3]; shs[P,M,3] or NULL; colors_precomp[P,3] or NULL (exactly one);
 *   opacities[P]; scales[P,2] + rotations[P,4] (w,x,y,z)  or  transMat_precomp[P,9] (exactly one);
 *   viewmatrix[16] = world_view_transform, projmatrix[16] = full_proj_transform, both as torch
 *   stores them (transposed, /root/reference/scene/cameras.py:56-58); cam_pos[3].
 * Outputs (caller-allocated): out_color[3,H,W], out_others[7,H,W] (channels: 0 sum w*depth, 1 alpha,
 *   2-4 view-space normal, 5 median depth, 6 distortion — contract pinned by
 *   /root/reference/gaussian_renderer/__init__.py:118-135), radii[P] (int32).
 * The three opaque buffers obtained through the callbacks must be kept by the caller and handed to
 * surfel_rasterize_backward unchanged.
 * Returns num_rendered (number of (tile, surfel) instances, >= 0) or a negative error code.
 */
int64_t surfel_rasterize_forward(
    surfel_alloc_fn geom_alloc, void* geom_user,
    surfel_alloc_fn binning_alloc, void* binning_user,
    surfel_alloc_fn image_alloc, void* image_user,
    int P, int D, int M,
    const float* background, int width, int height,
    const float* means3D, const float* shs, const float* colors_precomp, const float* opacities,
    const float* scales, float scale_modifier, const float* rotations, const float* transMat_precomp,
    const float* viewmatrix, const float* projmatrix, const float* cam_pos,
    float tan_fovx, float tan_fovy, int prefiltered,
    float* out_color, float* out_others, int* radii,
    int debug, void* stream);

/*
 * Backward.  Replaces `_C.rasterize_gaussians_backward` (reached from loss.backward(),
 * /root/reference/train.py:90).  R = value returned by the matching forward; geom/binning/image
 * buffers = the pointers the forward's callbacks returned.  dL_dout_color[3,H,W],
 * dL_dout_others[7,H,W].  All dL_d* outputs are caller-allocated; every element is written (zeros for culled
 * surfels), so they need NOT be zero-filled:
 *   dL_dmeans2D[P,3] (densification statistic consumed at /root/reference/scene/gaussian_model.py:405-407),
 *   dL_dnormal[P,3], dL_dopacity[P], dL_dcolors[P,3] (w.r.t. colors_precomp; in SH mode w.r.t. the SH colour BEFORE the
 *   forward's clamp_min(0), i.e. zero where the forward clamped), dL_dmeans3D[P,3], dL_dtransMat[P,9],
 *   dL_dsh[P,M,3] (may be NULL: skipped — callers that rebuild it from dL_dcolors, include/surfel_train.h), dL_dscales[P,2],
 *   dL_drots[P,4].  dL_dnormal and — unless transMat_precomp is given — dL_dtransMat are intermediates of the chain rule that no
 *   caller of the reference's Python API receives: either may be NULL and is then not written (saves 48 B/surfel of stores).
 * `scratch_alloc` provides the per-instance gradient records (R * 80 bytes + 8 bytes per tile + 1 byte per surfel; a record is written
 * at most once — on frames with >= 2^21 instances the records behind a tile's saturation point are never written and never read);
 * gradients are accumulated without atomics, so results are bit-reproducible run to run.
 */
int surfel_rasterize_backward(
    surfel_alloc_fn scratch_alloc, void* scratch_user,
    int P, int D, int M, int64_t R,
    const float* background, int width, int height,
    const float* means3D, const float* shs, const float* colors_precomp,
    const float* scales, float scale_modifier, const float* rotations, const float* transMat_precomp,
    const float* viewmatrix, const float* projmatrix, const float* cam_pos,
    float tan_fovx, float tan_fovy, const int* radii,
    const void* geom_buffer, const void* binning_buffer, const void* image_buffer,
    const float* dL_dout_color, const float* dL_dout_others,
    float* dL_dmeans2D, float* dL_dnormal, float* dL_dopacity, float* dL_dcolors,
    float* dL_dmeans3D, float* dL_dtransMat, float* dL_dsh, float* dL_dscales, float* dL_drots,
    int debug, void* stream);

/* Replaces `_C.mark_visible` (GaussianRasterizer.markVisible): present[P] (uint8) = view depth > 0.2. */
int surfel_mark_visible(int P, const float* means3D, const float* viewmatrix, const float* projmatrix,
                        uint8_t* present, void* stream);

/*
 * Replaces `simple_knn._C.distCUDA2` (/root/reference/scene/gaussian_model.py:20,134):
 * mean_dist2[P] = mean squared distance of every point to its 3 nearest neighbours.
 */
int surfel_knn_dist2(surfel_alloc_fn scratch_alloc, void* scratch_user, int P, const float* points,
                     float* mean_dist2, void* stream);

/* Introspection used by tests / bench: per-stage device timings (ms) of the last forward / backward
 * call made with debug != 0 on this thread; returns the number of stages written (<= cap). */
int surfel_last_stage_ms(float* ms, int cap);
int surfel_last_stage_ids(int* ids, int cap);
/* debug == 2 ("profile"): stages are bracketed with HIP events on `stream` WITHOUT synchronising;
 * this resolves every pending pair, adds the durations into sum_ms[stage] / count[stage]
 * (arrays of `cap` >= 11 entries, indexed by stage id) and returns the number of stage ids. */
int surfel_collect_stage_ms(float* sum_ms, int* count, int cap);
const char* surfel_stage_name(int stage);

/* Test entry: the library's stable LSD radix sort of (u32 key, u32 value) pairs on key bits [begin_bit, end_bit),
 * in place (device pointers, n < 2^30).  scratch_alloc is called once.  Returns 0 or a SURFEL_E_* code. */
int surfel_debug_sort_pairs(surfel_alloc_fn scratch_alloc, void* scratch_user, uint32_t* keys, uint32_t* vals, int64_t n,
                            int begin_bit, int end_bit, void* stream);

/* Process-wide defaults of the eight switches below; returns 0, or SURFEL_E_INVALID for an unknown name.  None of them changes a result
 * bit except where stated; each default is the measured best, the other settings exist for the tests that prove the
 * identity and for A / B runs of an unmodified caller (SURFEL_OPTIONS="name=value,..." in the environment of surfel_native.py).
 *   "cull"             1   exact footprint culling: tile emission restricted to the surfel's alpha >= 1/255 footprint, per-sub-tile
 *                          instance masks.  0: every (pixel, surfel) pair of the reference's tile rectangles is visited.
 *   "tile_depth_sort"  1   binning path — 2: surfel-order emission, then every tile orders its run by depth in LDS (small / medium frames);
 *                          0: depth-presorted emission (large frames); 1: by the previous frame's instances per tile.
 *   "capacity_binning" 1   per-tile-sort frames of <= 2^20 instances size their binning buffers from the largest count recent frames of that
 *                          size produced (+ 1/8) and never wait for this frame's count inside the forward; a frame that overflows is redone
 *                          with exact sizes (surfel_debug_last_binning() == 2).  The returned count is always exact.
 *   "large_sort"       2   sorts of > 2^20 items — 0: the library's three-launches-per-pass radix sort, 1: rocprim::radix_sort_pairs,
 *                          2: rocPRIM for key fields <= 16 bits and >= 4 M items, own passes otherwise.  Both stable.
 *   "tile_order"       0   which tile a blend workgroup takes — 1: XCD-contiguous runs, 2: longest lists first, dealt over the XCDs,
 *                          0: decided per frame on the device from the lists.  Scheduling only.
 *   "fwd_pipe"         1   blend forward with LDS-DMA double-buffered staging (0: the batch-synchronous kernel it is held bit-identical to).
 *   "tile_stream"      1   blend_fwd leaves, per list position it walked, the 80-B blend record and the 16 footprint bits in list order
 *                          (84 B x capacity of binning buffer, frames of <= 2^24 instances); blend_bwd stages from that contiguous stream
 *                          instead of surfel ids -> 112-B gather -> footprint test.  0: none is written, the backward gathers.
 *   "bwd_variant"      2   blend-backward walk — 0 per-row (every DPP row of 16 lanes = a 4x4-pixel sub-tile walks its own list),
 *                          1 per-quad (round 1's kernel: the reference the per-row walk is held bit-identical to), 3 scan (lanes are
 *                          instances, DPP row scans carry the per-pixel recurrences: deterministic, agrees with 0 / 1 to fp32 summation
 *                          noise, NOT bit for bit), 2 auto: 0 or 3 by a rule on the frame alone — 3 iff the frame holds >= 6 tile
 *                          instances per emitting surfel or 2^21 <= R < 2^26 instances (decided on the device where the host does not
 *                          know the count) — so the bits of a frame follow from the frame, never from timing or history.
 * Threading: the library keeps one pinned read-back buffer and one event per (host thread, device); calls are not re-entrant
 * per thread, and the intended layout is one process per GPU (torch.distributed.run).  Stage-timing events recorded with
 * debug >= 2 are kept until surfel_collect_stage_ms() (at most 8192 pairs; older ones are dropped). */
int surfel_set_option(const char* name, int value);

/* Multi-GPU overlap hook (process-wide; NULL removes it).  When set, surfel_rasterize_backward enqueues, after the blend
 * backward, a small kernel that finalises dL_dcolors (bit-identical to what the single-kernel path writes), calls
 * colour_ready(user) on the calling thread — the caller typically launches its all-gather of dL_dcolors there, ordered behind that
 * kernel on `stream` — and only then enqueues the per-surfel chain rule, which no longer touches dL_dcolors.  All other outputs
 * are unchanged.  Reference counterpart: none (the reference trains on one GPU); this serves the view-parallel exchange of
 * surfel_trainer.py. */
typedef void (*surfel_hook_fn)(void* user);
int surfel_set_backward_hook(surfel_hook_fn colour_ready, void* user);

/* Debug: a device buffer of 8 uint64 (caller-zeroed) that every following blend-backward launch accumulates into
 * — [0] lane slots issued (64 per wave visit), [1] lanes that held a composited (pixel, surfel) pair, [2] wave visits,
 * [3] (sub-tile | quad, instance) visits, [4] of those, the ones with at least one composited pair, [5] quad variant: 4x4 sub-tiles
 * with a composited pair — or NULL to switch the instrumented kernels off again. */
int surfel_debug_set_blend_stats(void* dev_u64x8);

/* Lazily counted frames.  surfel_rasterize_forward normally returns the exact number of tile instances — the one host wait of a
 * forward (the capacity path waits once everything is enqueued, but it waits).  With SURFEL_OPT_LAZY_COUNT in `debug`, a frame on the
 * capacity path returns at once with its CAPACITY (an upper bound: pass it on to surfel_rasterize_backward as num_rendered) and the
 * host runs ahead of the device; frames that take the exact path ignore the flag.  The caller owes the library one call of
 *     surfel_forward_count()
 * on the same thread before it lets the frame's results take effect (a trainer: after the backward, before the optimiser step):
 * it returns the exact count, or SURFEL_E_OVERFLOW if the frame held more instances than its capacity — its lists were truncated,
 * images and gradients are incomplete, and the caller renders the frame again (SURFEL_OPT_EXACT_BINNING) and recomputes what it
 * derived from it.  surfel_rasterize_backward may be called on such a frame BEFORE the count is collected (that is the point of
 * the flag): the frame's real total is on the device, every backward kernel compares it with num_rendered and returns at once when
 * the frame overflowed — nothing is read or written past the gradient records sized from the capacity, the gradient outputs are
 * left as they were.  Rare: the capacity is the largest count of the recent frames of that size plus 1/8.  A forward that finds the
 * previous lazy frame overflowed and unchecked fails with SURFEL_E_OVERFLOW instead of going on.  Without a pending lazy frame the
 * function returns the count of this thread's last forward.  Reference counterpart: num_rendered, the first return value of
 * rasterize_gaussians (diff-surfel-rasterization/rasterize_points.cu [UPSTREAM-RECALL]), which the reference reads back
 * synchronously. */
int64_t surfel_forward_count(void);

/* Debug: how the last forward of this thread sized its binning buffers — 0 exact (host wait for the count), 1 capacity,
 * 2 capacity overflowed and the frame was redone with exact sizes, 4 capacity with a lazily collected count (SURFEL_OPT_LAZY_COUNT). */
int surfel_debug_last_binning(void);

/* Debug: how often this host thread's per-frame-size history (binning capacity, tile-order verdicts; 16 sizes) had to drop a size to
 * make room for another one.  A dropped size costs its next frame the exact path and a host wait — speed only. */
int surfel_debug_capacity_evictions(void);

/* Debug: byte offsets inside the image buffer of a width x height frame under the current options (host arithmetic, no device needed):
 * out[0] total size, [1] final_T / M1 / M2 planes, [2] last / median contributor planes, [3] tile map.  The tile ranges start at
 * offset 0.  For the white-box tests and statistics scripts that read the buffer (diff_surfel_rasterization.image_layout mirrors it). */
int surfel_debug_image_layout(int width, int height, int64_t* out);

/* Debug / bench: what THIS GPU sustains, independent of the product's kernels (csrc/box_probe.hip) — 256 dependent empty launches, and a
 * grid of independent v_fma_f32 streams at 4 waves per SIMD timed with events on `stream` (the call synchronises).  scratch: >= 128 KiB of
 * device memory.  out[0] us per dependent launch boundary, [1] G wave-instructions / s of the FMA grid (whole chip, by events), [2] shader
 * clock that grid sustained in GHz (s_memtime ticks per 100 MHz s_memrealtime tick), [3] ms of the FMA grid, [4] shader cycles per
 * wave-instruction per SIMD over the grid's own span, [5] G wave-instructions / s over that span, [6] / [7] the cycles figure of [4] for grids
 * of v_add_f32 and of v_pk_fma_f32 (two FMAs per lane and instruction), [8] M wave-visits / s of a frozen stand-in for a blend-backward
 * visit (LDS reads + ~100 fp32 multiply-adds + transcendentals + selects + a 38-DPP reduction) at 4 workgroups per CU.  bench.py prints them as `box_probe` so that runs on different boxes of a pool can
 * be compared.  No reference counterpart. */
int surfel_debug_box_probe(void* scratch, int64_t scratch_bytes, float* out9, void* stream);
/* ... and its dependent-load latency: one lane chases a cycle of `hops` loads through `bytes` (>= 1 MiB) of `buf`; the footprint decides what
 * a hop hits (4 MiB: the XCD's L2, 1 GiB: HBM).  The call synchronises. */
int surfel_debug_latency_probe(void* buf, int64_t bytes, int hops, float* ns_per_hop, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SURFEL_HIP_H */
