// surfel_kernels.h — argument blocks and launchers shared between the kernel files and the C ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace surfel {

struct PreprocessArgs {
    int P, D, M, W, H, gx, gy;
    int cull;                 // 0: emit the full reference rect, footprints unbounded (test switch)
    int pack_tiles;           // 1 (P < 2^22): ident[i] = i | min(tiles_touched, 1023) << 22 — the depth-ordered scan then needs no gather by surfel id
    float scale_modifier;
    const float* means3D; const float* opacities; const float* scales; const float* rotations;
    const float* transMat_precomp; const float* colors_precomp; const float* shs;
    const float* viewmatrix; const float* projmatrix; const float* campos;
    float* rec; float* depths; uint32_t* depth_keys; uint32_t* ident; int* radii; uint32_t* tiles_touched; uint8_t* clamped;
    uint32_t* rects;          // packed emission rect per surfel (x0 | y0 << 10 | width << 20), a compact copy of record word 19
    float* shjac;             // [9][P] or NULL: d(SH colour c) / d(view direction q) of every visible surfel (plane 3 c + q) for preprocess_bwd
    uint32_t* total_instances;     // [2 * R_SLOTS], zeroed by the caller: partial sums of tiles_touched | of (tiles_touched > 0)
    uint32_t* block_totals;        // [workgroups] instances emitted by every workgroup's 256 surfels (bin_emit_kernel's scan), or NULL
    uint32_t* zero_a; uint32_t zero_a_words;   // scratch words this kernel clears for the launches that follow
    uint32_t* zero_b; uint32_t zero_b_words;   // (sort head, scan state) — saves two memset launches
    uint32_t* zero_c; uint32_t zero_c_words;   // capacity binning: head of the tile sort's scratch
};

struct BlendFwdArgs {
    int W, H, gx, gy;
    const uint2* ranges; const uint32_t* point_list; const float* rec; const float* bg;
    float* out_color; float* out_others; float* final_T; uint32_t* n_contrib;
    const int* tile_map; const uint32_t* map_flag; int map_len;      // tile of every workgroup where map_flag[0] != 0 (tile_order_kernel; -1: none), xcd_tile order otherwise; the grid size
    unsigned long long* stats;   // optional [8]: [6] += (pixel, surfel) pairs composited (surfel_debug_set_blend_stats)
    int avg_list;                // instances per tile where the host knows the count (exact binning path), else 0: picks the kernel (speed only)
    // the TILE STREAM (surfel_common.h): per list position the 80-B blend record + the 16 sub-tile footprint bits, in list order, for blend_bwd
    float4* strm_rec; uint32_t* strm_mask;      // [R][5] | [R], or NULL: none is written
    const uint32_t* totals; uint32_t* walk_word;      // [2 * R_SLOTS] partial sums written by preprocess (tile instances | visible surfels) -> the frame's backward walk (surfel_common.h: frame_walk)
};

struct BlendBwdArgs {
    int W, H, gx, gy;
    const uint2* ranges; const uint32_t* point_list; const float* rec; const float* bg;
    const float* final_T; const uint32_t* n_contrib;
    const int* tile_map; const uint32_t* map_flag; int map_len;      // as BlendFwdArgs
    const float* dL_dpix; const float* dL_dothers;
    float* grec;      // [R][GREC_F] per-instance gradient records: the records of a tile's list positions <= its cut are written exactly once, the rest never
    uint2* cut;       // [tiles] (depth bits, surfel index + 1) of the last instance of every tile that has a record, or NULL: every instance gets a record (surfel_blend_bwd.h: finish_tail)
    uint8_t* has_rec;      // with cut: [P] zeroed by the caller; set for every surfel that gets at least one record (preprocess_bwd skips the others)
    const float* depths;   // [P] view depths (the sort key's source)
    int variant;      // 0: per-DPP-row walk, 1: per-wave (8x8 quad) walk (0 / 1 bit-identical), 3: scan walk
    const float4* strm_rec; const uint32_t* strm_mask;      // the forward's tile stream (BlendFwdArgs), or NULL: the staging gathers the records by surfel id
    int scan_rule;    // 1: the scan kernel AND the rows / quad kernel selected by `variant` are launched; the device looks up which one runs (surfel_blend_bwd.h: device_picks_scan)
    const uint32_t* walk_word;   // the word blend_fwd left: WALK_ROWS / WALK_SCAN (surfel_common.h: frame_walk)
    const uint32_t* n_dev; uint32_t n_cap;      // capacity path: the frame's instance total on the device and the record capacity (= num_rendered).  n_dev[0] > n_cap:
                                                // a lazily counted frame that overflowed — its lists are truncated, the caller redoes it: every backward kernel returns at once
    unsigned long long* stats;   // optional [4]: lane slots issued, useful (pixel, surfel) lanes, wave visits, row / quad visits
};

struct PreprocessBwdArgs {
    int P, D, M, W, H;
    int coop;                 // 1: wave-cooperative gather of the instance gradient records (many records per surfel)
    int keep_colors;          // 1: dL_dcolors was already written by launch_colour_gradients (and may be on the wire): leave it alone
    const float* shjac;       // the forward's d(SH colour) / d(view direction) planes [9][P], or NULL: read the SH coefficients again
    float scale_modifier;
    const float* means3D; const int* radii; const float* shs; const uint8_t* clamped;
    const float* scales; const float* rotations; const float* transMat_precomp;
    const float* viewmatrix; const float* projmatrix; const float* campos;
    const float* rec; const uint32_t* tiles_touched; const float* grec;
    const uint2* cut; const uint8_t* has_rec; const float* depths; int gx;      // which of a surfel's instance records exist (BlendBwdArgs::cut); cut == NULL: all of them
    const uint32_t* n_dev; uint32_t n_cap;      // as BlendBwdArgs: an overflowed frame's first-instance slots point past the records — nothing is read, nothing written
    float* dL_dtransMat; float* dL_dnormal; float* dL_dopacity; float* dL_dcolors; float* dL_dsh;
    float* dL_dmeans2D; float* dL_dmeans3D; float* dL_dscales; float* dL_drots;
};

void launch_preprocess_fwd(const PreprocessArgs& a, hipStream_t s);
void launch_emit_instances(int P, float* rec, const uint32_t* rects, const uint32_t* order, uint32_t id_mask, const uint32_t* offsets_sorted, uint32_t* keys,
                           uint32_t* vals, int gx, uint32_t* zero_ptr, uint32_t zero_words, hipStream_t s);
void launch_tile_ranges(int64_t R, const uint32_t* keys, uint2* ranges, hipStream_t s);
int tile_map_len(int gx, int gy);
// force: 0 decide on the device, 1 always the XCD-contiguous order, 2 always longest-first round-robin
// verdict: optional device-visible word that receives 1 (uniform frame) / 2 (uneven lists)
void launch_tile_order(const uint2* ranges, int gx, int gy, int* map, uint32_t* map_flag /* zeroed */, int force, uint32_t* verdict, hipStream_t s);
void launch_blend_fwd(const BlendFwdArgs& a, hipStream_t s);
bool launch_blend_fwd_writes_stream(const BlendFwdArgs& a);      // does the kernel launch_blend_fwd picks for `a` write the tile stream (a.strm_rec given)?
void set_fwd_pipe(int v);             // 1 (default): software-pipelined staging (LDS-DMA of the next batch under the walk), 0: batch-synchronous kernel
void launch_mark_visible(int P, const float* means3D, const float* vm, uint8_t* present, hipStream_t s);
void launch_blend_bwd(const BlendBwdArgs& a, hipStream_t s);
void launch_blend_bwd_scan(const BlendBwdArgs& a, hipStream_t s);      // variant 3 (surfel_backward_scan.hip)
void launch_preprocess_bwd(const PreprocessBwdArgs& a, hipStream_t s);
// dL/dcolour alone (the sum of three floats of every gradient record, same order as preprocess_bwd -> the same bits), so that a
// caller can put it on the wire while preprocess_bwd still runs the geometry chain rule (surfel_set_backward_hook)
void launch_colour_gradients(const PreprocessBwdArgs& a, hipStream_t s);
void launch_knn(int P, const float* points, float* out, void* scratch, size_t scratch_bytes, hipStream_t s);
size_t knn_scratch_bytes(int P);
size_t radix_sort_scratch_bytes(size_t n);
void set_large_sort_impl(int v);      // for n > 2^20 — 0: three launches per pass (own), 1: rocprim::radix_sort_pairs, 2: auto (default)
int radix_sort_passes(size_t n, int begin_bit, int end_bit);
int radix_sort_result_buffer(size_t n, int begin_bit, int end_bit);
int radix_sort_pairs_u32(uint32_t* keys_a, uint32_t* vals_a, uint32_t* keys_b, uint32_t* vals_b, size_t n, int begin_bit, int end_bit,
                         void* scratch, hipStream_t s, bool head_zeroed = false);
size_t radix_sort_head_words(size_t n);       // words at the start of the sort scratch that must be zero (head_zeroed callers)
size_t scan_scratch_words(size_t n);          // zeroed scratch of launch_scan_gather
// decode: the ranges come from the sort's last pass (radix_sort_pairs_u32_devn) with complemented starts; the kernel turns them back
void launch_tile_depth_sort(int ntiles, int64_t R, uint2* ranges, uint32_t* point_list, const uint32_t* depth_keys, uint32_t* tmp_ids,
                            uint32_t* tmp_keys, uint32_t* tmp_rank, bool decode, hipStream_t s);
void launch_scan_gather(const uint32_t* vals, const uint32_t* order, int packed, uint32_t* out, size_t n, void* zeroed_scratch, hipStream_t s);
// capacity binning (surfel_sort.hip): scan + emission + tile-sort histograms in one launch, sort passes / ranges with the count on the device
bool capacity_binning_ok(size_t cap, int end_bit);
size_t capacity_sort_scratch_bytes(size_t cap, int end_bit);
size_t bin_emit_head_words();                 // words at the start of the capacity path's sort scratch that must be zero
void launch_bin_emit(int P, const uint32_t* tiles_touched, const uint32_t* rects, const uint32_t* block_totals, float* rec, uint32_t* keys, uint32_t* vals,
                     int gx, size_t cap, void* sort_scratch /* head zeroed */, int end_bit, uint32_t* n_out, uint32_t* n_host /* host-visible copy of the total, or NULL */,
                     hipStream_t s);
// ranges != NULL (zeroed, one uint2 per key value): the last pass writes [first, last + 1) of every key, first as its complement
int radix_sort_pairs_u32_devn(uint32_t* keys_a, uint32_t* vals_a, uint32_t* keys_b, uint32_t* vals_b, size_t cap, int end_bit, const uint32_t* n_dev,
                              void* scratch, uint2* ranges, hipStream_t s);
void launch_tile_ranges_devn(size_t cap, const uint32_t* n_dev, const uint32_t* keys, uint2* ranges, hipStream_t s);

}  // namespace surfel
