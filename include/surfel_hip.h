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
#define SURFEL_OPT_PBWD_COOP      (1 << 13)            /* backward: wave-cooperative gather of the gradient records (default: R >= 6 P and R >= 2^25) */
#define SURFEL_OPT_PBWD_THREAD    (1 << 14)            /* backward: per-thread gather of the gradient records */
#define SURFEL_OPT_EXACT_BINNING  (1 << 16)            /* forward: "capacity_binning" = 0 for this call (binning buffers sized after a host wait for the instance count) */
#define SURFEL_OPT_TILE_CUTS      (1 << 17)            /* backward: no gradient records behind a tile's saturation point, preprocess_bwd tests the tile cuts (default: R >= 2^21) */
#define SURFEL_OPT_ZERO_RECORDS   (1 << 18)            /* backward: zero records behind a tile's saturation point (default: R < 2^21); bit-identical to the cuts */
#define SURFEL_OPT_TILE_ORDER(m)  ((((m) + 1) & 3) << 19)  /* forward: "tile_order" = m (0, 1, 2) for this call (the matching backward follows the forward) */
#define SURFEL_OPT_LAZY_COUNT     (1 << 21)            /* forward: do not wait for the instance count (see surfel_forward_count) */
#define SURFEL_OPT_BWD_GATHER     (1 << 22)            /* backward: ignore the forward's tile stream ("tile_stream") and gather the records by surfel id; bit-identical */
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

/* Process-wide tuning / test switches; returns 0, or SURFEL_E_INVALID for an unknown name.
 *   "cull" (default 1): 0 disables every exact-preserving cull (tile emission restricted to the surfel's
 *          alpha>=1/255 footprint, per-quad / per-sub-tile instance masks) so the blend kernels visit every
 *          (pixel, surfel) pair of the reference's tile rectangles.  Results are bit-identical either way —
 *          that is what tests/test_gpu_parity.py::test_culling_is_exact checks.
 *   "tile_depth_sort" (default 1 = auto): binning path — 2: always order every tile's instance run by depth in LDS after an
 *          index-order emission (no P-sized depth sort; fastest for small / medium frames), 0: always depth-presort the surfels
 *          (large frames), 1: choose by the previous frame's instances per tile.  Results are bit-identical
 *          (tests/test_gpu_parity.py::test_binning_paths_are_identical).
 *   "large_sort": sorts of more than 2^20 items (the P-sized depth sort and the R-sized tile sort of large frames) — 0: the library's
 *          three-launches-per-pass radix sort, 1: rocprim::radix_sort_pairs, 2 (default): rocPRIM for key fields of <= 16 bits and for
 *          >= 4 M items, the library's passes otherwise (profiles/r02_large_sort.md).  Both stable: identical results
 *          (tests/test_gpu_parity.py::test_radix_sort_is_stable_and_exact runs both).
 *   "bwd_variant" (default 2 = auto): blend-backward walk — 0: every DPP row of 16 lanes (a 4x4-pixel sub-tile) walks its own
 *          instance list, row totals gathered through private LDS slots; 1: every wave (8x8 pixels) walks one list (round 1's
 *          kernel).  Same per-pair arithmetic and the same summation tree: gradients are bit-identical
 *          (tests/test_gpu_parity.py::test_backward_variants_are_identical), so the choice between these two only ever changes
 *          speed.  Under auto ("scan_large", default 1) the scan walk (3) is launched beside the rows / quad kernel and the DEVICE
 *          decides from the frame's totals which one runs: frames with >= 6 tile instances per emitting surfel (trained / wide-footprint
 *          frames: -13 % / -5 %) or with 2^21 <= R < 2^26 tile instances (-7 % at 8 M instances) take walk 3, the other kernel returns at
 *          once — a rule on the frame, so the bits of a frame follow from the frame alone.
 *          3: the scan walk (surfel_backward_scan.hip: lanes are instances, DPP row scans carry the per-pixel recurrences, gradients
 *          accumulate in registers) — deterministic, but a different summation order: agrees with rows / quad to fp32 summation noise,
 *          not bit for bit; 7 % faster than both on frames of several million instances, at parity around 2 M, 11 % slower on
 *          0.5 M instances of small random footprints (profiles/r03_blend_bwd_scan.md).  4: auto over all three walks by the same timed probes — fastest, but which bits a frame
 *          gets then depends on the probes' verdict.
 *   "bwd_tune" (default 1): how auto chooses.  1: per (device, width, height, octave of tile instances per surfel), two backward
 *          calls in every 32 are timed with HIP events on the launch stream (one per walk; polled later, never synchronised) and
 *          the faster walk per tile instance is launched alone in between; 0: both kernels are launched every call and the device decides from the frame's
 *          totals (rows iff tile instances <= 4 x emitting surfels) — also what happens before both walks have been timed and
 *          while the stream is being captured into a graph.
 *   "tile_order" (default 0): which tile every blend workgroup takes.  1: workgroup b -> XCD b % 8 walks a contiguous run of tiles
 *          (neighbouring tiles share surfel records in that XCD's L2) — right for frames whose tiles hold similar lists; 2: groups of
 *          4 adjacent tiles, longest lists first, dealt round-robin over the XCDs — right for object-centred / trained frames, where a
 *          few hundred tiles hold the long lists (the trained leg's blend kernels ran at 0.25 of the VALU issue peak
 *          with order 1: the XCDs owning the image's middle rows did the work, the heaviest tiles finished alone); 0: decided per
 *          frame on the device from the lists (busiest XCD > 1.15x its share, or a list > 4 average lists -> 2).  Scheduling only:
 *          results are bit-identical (tests/test_gpu_parity.py::test_tile_order_is_scheduling_only).
 *   "fwd_pipe" (default 1): blend forward kernel — 1: software-pipelined staging (LDS-DMA of the next batch's whole records under the
 *          walk, per-row byte lists, LDS prefetch of the next visit; DESIGN.md section 4 "Round 4"), 0: round 3's batch-synchronous kernel.
 *          Same walk, same per-pair arithmetic: bit-identical outputs (tests/test_gpu_parity.py::test_forward_kernels_are_identical).
 *   "fat_sort" (default 1): look-back sort passes over <= 2^20 items use 8192-item tiles staged through LDS (0: 2048-item tiles);
 *          "host_total" (default 1): capacity-path frames store their instance total into mapped pinned memory from the emission
 *          kernel (0: a device-to-host copy in the stream).  Speed only (tests/test_gpu_parity.py::test_radix_sort_is_stable_and_exact
 *          runs both tile sizes).
 *   "pbwd_coop" (default -1): record gather of the per-surfel backward — -1: by rule (wave-cooperative where R >= 6 P and R >= 2^25,
 *          per thread otherwise; measured again in round 3: the cooperative form loses 15 % at C2H and 25 % at C4), 0 / 1: forced.
 *          Bit-identical either way (tests/test_gpu_parity.py::test_record_gather_variants_are_identical).
 *   "capacity_binning" (default 1): frames on the per-tile-depth-sort path with <= 2^20 tile instances size their binning buffers
 *          from the largest instance count recent frames of the same size produced (+ 1/8 head room) instead of waiting for this
 *          frame's count in the middle of the forward: scan, emission and the tile sort's histograms run as ONE kernel right behind
 *          preprocess, the sort passes and the tile ranges read the count from the device, and the host looks at the count only once
 *          the whole forward is enqueued.  A frame that overflows its capacity is redone with exact sizes (surfel_debug_last_binning
 *          reports 2).  Results are bit-identical either way (tests/test_gpu_parity.py::test_capacity_binning_is_identical); the
 *          returned instance count is always the exact one.
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
 * grid of independent v_fma_f32 streams at 8 waves per SIMD timed with events on `stream` (the call synchronises).  scratch: >= 128 KiB of
 * device memory.  out[0] us per dependent launch boundary, [1] G wave-instructions / s of the FMA grid (whole chip, by events), [2] shader
 * clock that grid sustained in GHz (s_memtime ticks per 100 MHz s_memrealtime tick), [3] ms of the FMA grid, [4] shader cycles per
 * wave-instruction per SIMD over the grid's own span (nominal 2), [5] G wave-instructions / s over that span.  bench.py prints them as `box_probe` so that runs on different boxes of a pool can
 * be compared.  No reference counterpart. */
int surfel_debug_box_probe(void* scratch, int64_t scratch_bytes, float* out6, void* stream);

/* Debug: the walk the "bwd_tune" probes currently favour for frames of this size on the current device (the most used entry of
 * that size) — 0 per-row, 1 per-quad, -1 not decided yet (fewer than two timed calls have completed). */
int surfel_debug_walk_choice(int width, int height);

#ifdef __cplusplus
}
#endif
#endif /* SURFEL_HIP_H */
