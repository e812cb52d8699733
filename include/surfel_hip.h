/*
 * surfel_hip.h — C ABI of libsurfel_hip.so: the MI355X (gfx950) differentiable surfel rasterizer and the simple-knn kernel.
 *
 * Drop-in boundary for the hot path of hbb1/2d-gaussian-splatting.  The reference binds the same functionality through two
 * pybind11 modules that are absent (un-vendored submodules, /root/reference/.gitmodules:1-6):
 *     diff_surfel_rasterization._C : rasterize_gaussians / rasterize_gaussians_backward / mark_visible
 *     simple_knn._C                : distCUDA2
 * Plain pointers and sizes only — no torch types; every pointer is a DEVICE pointer unless stated; `stream` is a hipStream_t
 * passed as void* (NULL = default stream).  The library never allocates: scratch memory comes from caller-supplied allocator
 * callbacks (the reference's std::function<char*(size_t)> resize functors), so PyTorch's caching allocator owns all memory.
 * Errors: >= 0 on success, a negative SURFEL_E_* code on failure; surfel_last_error() returns a thread-local message.
 * Threading: one pinned read-back buffer and one event per (host thread, device); calls are not re-entrant per thread; intended
 * layout: one process per GPU.  Forward -> backward state travels in the three buffers of the call, so any number of forwards may
 * sit between a forward and its backward.  Diagnostics (stage timings, probes, counters): include/surfel_debug.h.
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
#define SURFEL_E_OVERFLOW (-5)  /* a lazily counted frame held more instances than its capacity: render it again */
/* `debug` argument of the rasterizer entry points: low byte = mode (0 off; 1 synchronise + check after every stage, the reference's
 * `debug` flag /root/reference/gaussian_renderer/__init__.py:49; 2 / 3 record stage-timing events), upper bits = PER-CALL overrides of
 * the process-wide options below.  Every override leaves results bit-identical except SURFEL_OPT_BWD_SCAN (summation order). */
#define SURFEL_OPT_NO_CULL        (1 << 8)             /* forward: "cull" = 0 */
#define SURFEL_OPT_TILE_SORT(m)   ((((m) + 1) & 3) << 9)   /* forward: "tile_depth_sort" = m */
#define SURFEL_OPT_BWD_QUAD       (1 << 11)            /* backward: "bwd_variant" = 1 */
#define SURFEL_OPT_BWD_ROWS       (1 << 12)            /* backward: "bwd_variant" = 0 */
#define SURFEL_OPT_PBWD_COOP      (1 << 13)            /* backward: wave-cooperative gather of the gradient records (default: R >= 6 P and R >= 2^25) */
#define SURFEL_OPT_PBWD_THREAD    (1 << 14)            /* backward: per-thread gather of the gradient records */
#define SURFEL_OPT_BWD_SCAN       (1 << 15)            /* backward: "bwd_variant" = 3 */
#define SURFEL_OPT_EXACT_BINNING  (1 << 16)            /* forward: "capacity_binning" = 0 */
#define SURFEL_OPT_TILE_CUTS      (1 << 17)            /* backward: no gradient records behind a tile's saturation point (default: R >= 2^21) */
#define SURFEL_OPT_ZERO_RECORDS   (1 << 18)            /* backward: zero records there instead (default: R < 2^21) */
#define SURFEL_OPT_TILE_ORDER(m)  ((((m) + 1) & 3) << 19)  /* forward: "tile_order" = m */
#define SURFEL_OPT_LAZY_COUNT     (1 << 21)            /* forward: do not wait for the instance count (surfel_forward_count) */
#define SURFEL_OPT_BWD_GATHER     (1 << 22)            /* backward: ignore the forward's tile stream, gather the records by surfel id */
#define SURFEL_OPT_NO_STREAM      (1 << 23)            /* forward: no backward will follow (inference, no_grad): leave no tile stream behind */
#define SURFEL_OPT_PBWD_NO_JAC    (1 << 24)            /* backward: read the SH block again instead of the forward's d(colour)/d(direction) rows */
/* Allocator callback: `bytes` bytes of device memory, 256-byte aligned, valid until the caller frees it (SURVEY.md 8b "ownership"). */
typedef void* (*surfel_alloc_fn)(void* user, size_t bytes);

int surfel_abi_version(void);
const char* surfel_last_error(void);
/* Forward.  Replaces `_C.rasterize_gaussians` (call site /root/reference/gaussian_renderer/__init__.py:97-106).
 *   P surfels, D active SH degree (0..3), M SH coefficients stored per surfel (16).  background[3]; means3D[P,3]; shs[P,M,3] or NULL;
 *   colors_precomp[P,3] or NULL (exactly one); opacities[P]; scales[P,2] + rotations[P,4] (w,x,y,z) or transMat_precomp[P,9] (exactly
 *   one); viewmatrix[16] = world_view_transform, projmatrix[16] = full_proj_transform as torch stores them (transposed,
 *   /root/reference/scene/cameras.py:56-58); cam_pos[3].
 * Outputs (caller-allocated): out_color[3,H,W], out_others[7,H,W] (0 sum w*depth, 1 alpha, 2-4 view-space normal, 5 median depth,
 *   6 distortion: /root/reference/gaussian_renderer/__init__.py:118-135), radii[P] int32.  The three buffers obtained through the
 *   callbacks are kept by the caller and handed to surfel_rasterize_backward unchanged.
 * Returns num_rendered ((tile, surfel) instances, >= 0) or a negative error code. */
int64_t surfel_rasterize_forward(
    surfel_alloc_fn geom_alloc, void* geom_user, surfel_alloc_fn binning_alloc, void* binning_user, surfel_alloc_fn image_alloc, void* image_user,
    int P, int D, int M, const float* background, int width, int height,
    const float* means3D, const float* shs, const float* colors_precomp, const float* opacities,
    const float* scales, float scale_modifier, const float* rotations, const float* transMat_precomp,
    const float* viewmatrix, const float* projmatrix, const float* cam_pos, float tan_fovx, float tan_fovy, int prefiltered,
    float* out_color, float* out_others, int* radii, int debug, void* stream);
/* Backward.  Replaces `_C.rasterize_gaussians_backward` (reached from loss.backward(), /root/reference/train.py:90).  R = the matching
 * forward's return value; dL_dout_color[3,H,W], dL_dout_others[7,H,W].  Every element of every output is written (zeros for culled
 * surfels): dL_dmeans2D[P,3] (the densification statistic of /root/reference/scene/gaussian_model.py:405-407), dL_dopacity[P],
 * dL_dcolors[P,3] (in SH mode w.r.t. the SH colour before the clamp: zero where the forward clamped), dL_dmeans3D[P,3], dL_dscales[P,2],
 * dL_drots[P,4]; dL_dsh[P,M,3], dL_dnormal[P,3] and — unless transMat_precomp is given — dL_dtransMat[P,9] may be NULL (not written).
 * `scratch_alloc` provides the per-instance gradient records (R * 80 B + 8 B per tile + 1 B per surfel); gradients are accumulated
 * without atomics: bit-reproducible run to run. */
int surfel_rasterize_backward(
    surfel_alloc_fn scratch_alloc, void* scratch_user, int P, int D, int M, int64_t R, const float* background, int width, int height,
    const float* means3D, const float* shs, const float* colors_precomp, const float* scales, float scale_modifier, const float* rotations,
    const float* transMat_precomp, const float* viewmatrix, const float* projmatrix, const float* cam_pos, float tan_fovx, float tan_fovy,
    const int* radii, const void* geom_buffer, const void* binning_buffer, const void* image_buffer,
    const float* dL_dout_color, const float* dL_dout_others, float* dL_dmeans2D, float* dL_dnormal, float* dL_dopacity, float* dL_dcolors,
    float* dL_dmeans3D, float* dL_dtransMat, float* dL_dsh, float* dL_dscales, float* dL_drots, int debug, void* stream);
/* Replaces `_C.mark_visible` (GaussianRasterizer.markVisible): present[P] (uint8) = view depth > 0.2. */
int surfel_mark_visible(int P, const float* means3D, const float* viewmatrix, const float* projmatrix, uint8_t* present, void* stream);
/* Replaces `simple_knn._C.distCUDA2` (/root/reference/scene/gaussian_model.py:20,134): mean squared distance to the 3 nearest neighbours. */
int surfel_knn_dist2(surfel_alloc_fn scratch_alloc, void* scratch_user, int P, const float* points, float* mean_dist2, void* stream);
/* Process-wide defaults of the eight switches below; returns 0, or SURFEL_E_INVALID for an unknown name.  None changes a result bit except
 * where stated; each default is the measured best; the other settings serve the identity tests and A / B runs.
 *   "cull"             1   exact footprint culling (tile emission and per-sub-tile masks restricted to the alpha >= 1/255 footprint); 0: every pair of the reference's tile rectangles
 *   "tile_depth_sort"  1   binning path: 2 surfel-order emission + per-tile depth sort in LDS, 0 depth-presorted emission, 1 by the previous frame's instances per tile
 *   "capacity_binning" 1   frames of <= 2^20 instances size their binning buffers from recent frames' counts and never wait for this frame's count; an overflowing frame is redone
 *   "large_sort"       2   sorts of > 2^20 items: 0 own three-launch passes, 1 rocprim::radix_sort_pairs, 2 rocPRIM above 2^25 items
 *   "tile_order"       0   which tile a blend workgroup takes: 1 XCD-contiguous runs, 2 longest lists first over the XCDs, 0 decided per frame on the device
 *   "fwd_pipe"         1   blend forward with LDS-DMA double-buffered staging (0: the batch-synchronous kernel, same bits)
 *   "tile_stream"      1   blend_fwd leaves the walked records + footprint bits in list order (84 B x binning capacity, frames <= 2^24 instances) for blend_bwd's staging
 *   "bwd_variant"      2   blend-backward walk: 0 per-row, 1 per-quad (bit-identical to 0), 3 scan (deterministic; agrees to fp32 summation noise), 2 = 0 or 3 by a rule on the frame alone
 */
int surfel_set_option(const char* name, int value);

/* Multi-GPU overlap hook (process-wide; NULL removes it): surfel_rasterize_backward finalises dL_dcolors right behind the blend backward,
 * calls colour_ready(user) on the calling thread (the caller launches its all-gather of dL_dcolors there), then enqueues the per-surfel
 * chain rule.  All outputs unchanged.  No reference counterpart (the reference trains on one GPU). */
typedef void (*surfel_hook_fn)(void* user);
int surfel_set_backward_hook(surfel_hook_fn colour_ready, void* user);

/* Lazily counted frames (SURFEL_OPT_LAZY_COUNT): a capacity-path forward returns at once with its CAPACITY (pass it to the backward as R);
 * the caller owes one surfel_forward_count() on the same thread before the frame's results take effect: the exact count, or
 * SURFEL_E_OVERFLOW (lists truncated: render the frame again with SURFEL_OPT_EXACT_BINNING).  The backward may run before the count is
 * collected: its kernels compare the device-side total with R and return at once on an overflowed frame.  Reference counterpart:
 * num_rendered, read back synchronously by rasterize_gaussians [UPSTREAM-RECALL]. */
int64_t surfel_forward_count(void);

#ifdef __cplusplus
}
#endif
#endif /* SURFEL_HIP_H */
