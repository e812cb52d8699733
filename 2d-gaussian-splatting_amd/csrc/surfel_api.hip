// surfel_api.hip — C ABI (include/surfel_hip.h) of libsurfel_hip.so: stage orchestration, scratch
// carving.  Host code only; kernels live in surfel_forward.hip,
// surfel_backward.hip and knn.hip.
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include <hip/hip_runtime.h>

#include "../../include/surfel_debug.h"
#include "surfel_common.h"
#include "surfel_kernels.h"
#include "train_kernels.h"

using namespace surfel;

namespace {

thread_local std::string g_err;
int g_opt_cull = 1;        // surfel_set_option("cull", .)
int g_opt_tile_sort = 1;   // surfel_set_option("tile_depth_sort", .): 0 never, 1 auto (by last frame's R / tiles), 2 always
int g_opt_bwd_variant = 2; // surfel_set_option("bwd_variant", .): 0 per-row walk, 1 per-quad walk (0 / 1 bit-identical), 2 auto = rows or scan by the device rule, 3 scan walk
surfel_hook_fn g_colour_hook = nullptr;   // surfel_set_backward_hook
void* g_colour_hook_user = nullptr;
int g_opt_stream = 1;      // surfel_set_option("tile_stream", .): blend_fwd leaves the tile stream for blend_bwd (surfel_common.h); 0: blend_bwd gathers
unsigned long long* g_blend_stats = nullptr;   // surfel_debug_set_blend_stats
thread_local int64_t g_last_R = -1; thread_local int g_last_W = 0, g_last_H = 0;   // auto heuristic (speed only; results identical; a stale
                                                                                    // value from another device / stream only costs one slower frame)
// capacity binning: the largest instance count recent frames of a size produced (per host thread; speed only)
// (16 entries: a COLMAP capture mixes a handful of resolutions; an evicted size falls back to the exact path + a host wait for one
// frame — counted, surfel_debug_capacity_evictions.  boost: head room doubles (1/8 -> 1/4 -> 1/2) while frames of the size keep
// overflowing — views of one capture differ 2-3x in instance count — and decays after 256 frames without an overflow.)
struct CapEntry { int W = 0, H = 0; int64_t maxR = -1; unsigned frames = 0; int uniform_streak = 0; int boost = 0; unsigned last_overflow = 0; };
constexpr int kCapEntries = 16;
thread_local CapEntry g_caps[kCapEntries];
thread_local unsigned g_cap_next = 0;
thread_local int g_cap_evictions = 0;
int g_opt_capacity = 1;    // surfel_set_option("capacity_binning", .)
int g_opt_tile_order = 0;  // surfel_set_option("tile_order", .): 0 decided per frame on the device, 1 XCD-contiguous runs, 2 longest lists first
thread_local int g_last_binning = 0;      // 0 exact-size path, 1 capacity path, 2 capacity path overflowed and the frame was redone (surfel_debug_last_binning)
CapEntry* cap_entry(int W, int H, bool create) {
    for (auto& c : g_caps) if (c.W == W && c.H == H) return &c;
    if (!create) return nullptr;
    CapEntry* c = &g_caps[g_cap_next++ % kCapEntries];
    if (c->W != 0) g_cap_evictions++;
    *c = CapEntry{};
    c->W = W; c->H = H;
    c->frames = ~0u;      // (marks "just created": the caller clears the slot's pinned tile-order verdict, run_tile_order)
    return c;
}
thread_local float g_stage_ms[16];
thread_local int g_stage_n = 0;
thread_local int g_stage_id[16];

const char* kStageNames[] = {"preprocess_fwd", "depth_sort_scan", "emit_instances", "tile_sort", "tile_ranges", "blend_fwd",
                             "zero_grec", "blend_bwd", "preprocess_bwd", "knn", "tile_depth_sort"};
enum Stage { ST_PRE = 0, ST_SCAN, ST_EMIT, ST_SORT, ST_RANGES, ST_BLEND, ST_ZERO, ST_BBWD, ST_PBWD, ST_KNN, ST_TSORT };

int fail(int code, const char* what, hipError_t e = hipSuccess) {
    g_err = what;
    if (e != hipSuccess) { g_err += ": "; g_err += hipGetErrorString(e); }
    return code;
}

#define HIP_TRY(expr)                                                    \
    do {                                                                 \
        hipError_t _e = (expr);                                          \
        if (_e != hipSuccess) return fail(SURFEL_E_HIP, #expr, _e);      \
    } while (0)

inline size_t align_up(size_t v, size_t a = 256) { return (v + a - 1) / a * a; }

// Bump carving of an opaque buffer; with base == nullptr it only measures.
struct Carver {
    char* base; size_t off = 0;
    explicit Carver(void* b) : base(static_cast<char*>(b)) {}
    template <typename T> T* take(size_t count) {
        off = align_up(off);
        T* p = base ? reinterpret_cast<T*>(base + off) : nullptr;
        off += count * sizeof(T);
        return p;
    }
    size_t size() const { return align_up(off); }
};

struct GeomState {   // per-surfel state ("geomBuffer")
    float* rec; float* depths; uint32_t* tiles_touched; uint8_t* clamped;
    uint32_t *dkey_a, *dkey_b, *ord_a, *ord_b;   // depth-bit keys / surfel order (double buffers of the P-sized sort)
    uint32_t* offsets;                            // inclusive scan of tiles_touched in depth order
    uint32_t* rects;                              // packed tile rect of every surfel (copy of record word 19: emission reads 4 B instead of a record line)
    float* shjac;                                 // [9][P]: d(SH colour) / d(view direction) planes for preprocess_bwd (written unless SURFEL_OPT_NO_STREAM)
    char* temp; size_t temp_bytes;                // scratch of the P-sized sort, then of the scan
    static GeomState carve(void* base, int P, size_t temp_bytes, size_t* total) {
        Carver c(base); GeomState g;
        g.rec = c.take<float>((size_t)P * REC_F);
        g.depths = c.take<float>(P);           // float32 view depth per surfel (sort key source; kept for inspection)
        g.tiles_touched = c.take<uint32_t>(P);
        g.clamped = c.take<uint8_t>(P);
        g.dkey_a = c.take<uint32_t>(P); g.dkey_b = c.take<uint32_t>(P);
        g.ord_a = c.take<uint32_t>(P); g.ord_b = c.take<uint32_t>(P);
        g.offsets = c.take<uint32_t>(P);
        g.rects = c.take<uint32_t>(P);
        g.shjac = c.take<float>((size_t)9 * P);
        g.temp = c.take<char>(temp_bytes);
        g.temp_bytes = temp_bytes;
        if (total) *total = c.size();
        return g;
    }
};

struct BinState {    // per-instance state ("binningBuffer"); point_list is always at a fixed offset
    uint32_t* point_list; uint32_t* vals_alt; uint32_t* keys_a; uint32_t* keys_b; char* sort_temp; size_t sort_temp_bytes;
    float4* strm_rec = nullptr; uint32_t* strm_mask = nullptr;      // the tile stream (surfel_common.h), or NULL
    static bool wants_stream(size_t R) { return g_opt_stream != 0 && R > 0 && (long long)R <= STRM_MAX_R; }
    // stream: the forward's decision (wants_stream of ITS capacity); the backward never carves it — it is called with the exact count
    // where the forward carved with a capacity, and finds the stream through the registry below
    static BinState carve(void* base, size_t R, size_t sort_bytes, size_t* total, bool stream = false) {
        Carver c(base); BinState b;
        b.point_list = c.take<uint32_t>(R);
        b.vals_alt = c.take<uint32_t>(R);
        b.keys_a = c.take<uint32_t>(R);
        b.keys_b = c.take<uint32_t>(R);
        if (stream) {
            b.strm_rec = c.take<float4>(R * STRM_Q);
            b.strm_mask = c.take<uint32_t>(R);
        }
        b.sort_temp = c.take<char>(sort_bytes);
        b.sort_temp_bytes = sort_bytes;
        if (total) *total = c.size();
        return b;
    }
};

struct ImgState {    // per-pixel / per-tile state ("imgBuffer")
    uint2* ranges; uint32_t* total; float* final_T; uint32_t* n_contrib; int* tile_map;
    static ImgState carve(void* base, int W, int H, size_t* total) {
        Carver c(base); ImgState im;
        const size_t tiles = (size_t)((W + TILE - 1) / TILE) * ((H + TILE - 1) / TILE);
        // [tiles] ranges + R_SLOTS partial instance totals + R_SLOTS partial visible-surfel counts + the instance total of the capacity
        // path | the tile-map flag | the frame's backward walk (blend_fwd) | a spare word (zeroed together)
        im.ranges = c.take<uint2>(tiles + R_SLOTS + 2);
        im.total = reinterpret_cast<uint32_t*>(im.ranges + tiles);
        im.final_T = c.take<float>((size_t)3 * W * H);
        im.n_contrib = c.take<uint32_t>((size_t)2 * W * H);
        im.tile_map = c.take<int>((size_t)tile_map_len((W + TILE - 1) / TILE, (H + TILE - 1) / TILE));      // blend workgroup -> tile (tile_order_kernel)
        if (total) *total = c.size();
        return im;
    }
};

// Which binning buffers hold a tile stream (surfel_common.h), and where.  The forward decides (option, capacity, blend kernel) and
// registers every binning buffer it fills — with a stream or without; the backward, handed the same buffer later (possibly on autograd's
// worker thread, possibly after other forwards), looks it up and stages from the stream if it finds one.  A buffer the registry no
// longer knows (more than kStreamRegs forwards in between) is gathered by surfel id: always valid, same bits.  An address can only
// be re-registered after its previous frame's buffers were freed, i.e. when no backward of that frame can come any more.
struct StreamReg { const void* bin = nullptr; const void* img = nullptr; const float4* rec = nullptr; const uint32_t* mask = nullptr; bool jac = false; };
constexpr int kStreamRegs = 64;
StreamReg g_stream_regs[kStreamRegs];
unsigned g_stream_next = 0;
std::mutex g_stream_mu;
// (keyed by the frame's binning AND image buffer: a caller that restores a saved binning buffer at an address still registered for
// another, already freed frame hands the backward a different image buffer with it — no match, the backward gathers)
// jac: the forward also left the d(SH colour) / d(direction) rows in the geometry buffer (preprocess_bwd reads them instead of the SH block)
void stream_register(const void* bin, const void* img, const float4* rec, const uint32_t* mask, bool jac) {
    std::lock_guard<std::mutex> lk(g_stream_mu);
    for (auto& r : g_stream_regs) if (r.bin == bin) { r.img = img; r.rec = rec; r.mask = mask; r.jac = jac; return; }
    g_stream_regs[g_stream_next++ % kStreamRegs] = StreamReg{bin, img, rec, mask, jac};
}
bool stream_lookup(const void* bin, const void* img, const float4** rec, const uint32_t** mask, bool* jac) {
    std::lock_guard<std::mutex> lk(g_stream_mu);
    for (auto& r : g_stream_regs) if (r.bin == bin && r.img == img) {
        *jac = r.jac;
        if (!r.rec) return false;
        *rec = r.rec; *mask = r.mask;
        return true;
    }
    return false;
}

// Stage timing on the caller's stream with HIP events.
//   mode 1 (debug): synchronise + hipGetLastError after every stage (the reference's `debug` flag).
//   mode 2 (profile): record events only; durations are resolved later by surfel_collect_stage_ms()
//                     so a timed region is not perturbed by host synchronisation.
struct PendingStage { int stage; hipEvent_t e0, e1; };
std::vector<PendingStage> g_pending;   // process-wide: autograd runs backward on its own thread
constexpr size_t kMaxPending = 8192;   // debug >= 2 without surfel_collect_stage_ms(): bounded (oldest timings are dropped)
std::mutex g_pending_mu;

struct StageTimer {
    int mode; hipStream_t s; hipEvent_t e0 = nullptr, e1 = nullptr;
    StageTimer(int mode_, hipStream_t s_) : mode(mode_), s(s_) {
        if (mode == 1) { (void)hipEventCreate(&e0); (void)hipEventCreate(&e1); }
    }
    ~StageTimer() { if (mode == 1) { (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); } }
    bool armed = false;
    void begin(int stage = -1) {
        armed = mode == 1 || mode == 2 || (mode == 3 && stage == ST_BBWD);
        if (!armed) return;
        if (mode >= 2) { (void)hipEventCreate(&e0); (void)hipEventCreate(&e1); }
        (void)hipEventRecord(e0, s);
    }
    int end(int stage) {   // returns hip error as int
        if (!armed) return 0;
        armed = false;
        (void)hipEventRecord(e1, s);
        if (mode >= 2) {
            std::lock_guard<std::mutex> lk(g_pending_mu);
            if (g_pending.size() >= kMaxPending) {       // nobody collects: drop the oldest pair instead of growing without bound
                (void)hipEventDestroy(g_pending.front().e0); (void)hipEventDestroy(g_pending.front().e1);
                g_pending.erase(g_pending.begin());
            }
            g_pending.push_back({stage, e0, e1});
            return 0;
        }
        hipError_t e = hipEventSynchronize(e1);
        if (e != hipSuccess) return (int)e;
        e = hipGetLastError();
        if (e != hipSuccess) return (int)e;
        float ms = 0.f; (void)hipEventElapsedTime(&ms, e0, e1);
        if (g_stage_n < 16) { g_stage_ms[g_stage_n] = ms; g_stage_id[g_stage_n] = stage; g_stage_n++; }
        return 0;
    }
};
#define STAGE_END(timer, st)                                                               \
    do {                                                                                   \
        int _e = (timer).end(st);                                                          \
        if (_e) return fail(SURFEL_E_HIP, kStageNames[st], (hipError_t)_e);                \
    } while (0)

// One event + one pinned read-back buffer per (host thread, device): a thread that rasterizes on a second GPU gets its own pair
// instead of recording an event created on another device.
constexpr int kMaxDevices = 32;
constexpr int kPinnedTotal = R_SLOTS + kCapEntries;      // word of the pinned buffer that receives a capacity-path frame's instance total
struct PerDevice { hipEvent_t ev = nullptr; uint32_t* pinned = nullptr; };
PerDevice* per_device(int dev = -1) {      // dev < 0: the calling thread's current device
    thread_local PerDevice tab[kMaxDevices];
    int d = dev;
    if (d < 0 && hipGetDevice(&d) != hipSuccess) return nullptr;
    if (d < 0 || d >= kMaxDevices) return nullptr;
    return &tab[d];
}

hipEvent_t r_event() {
    PerDevice* pd = per_device();
    if (!pd) return nullptr;
    if (!pd->ev) { if (hipEventCreateWithFlags(&pd->ev, hipEventDisableTiming) != hipSuccess) pd->ev = nullptr; }
    return pd->ev;
}

uint32_t* pinned_u32() {
    PerDevice* pd = per_device();
    if (!pd) return nullptr;
    if (!pd->pinned) {
        // R_SLOTS partial instance totals (D2H copy target) + one tile-order verdict word per capacity-table entry (written by the device)
        // + the instance total of a capacity-path frame (written by bin_emit_kernel: kPinnedTotal)
        if (hipHostMalloc(reinterpret_cast<void**>(&pd->pinned), sizeof(uint32_t) * (R_SLOTS + kCapEntries + 8), hipHostMallocMapped) != hipSuccess) pd->pinned = nullptr;
        else for (int k = 0; k < R_SLOTS + kCapEntries + 8; k++) pd->pinned[k] = 0u;
    }
    return pd->pinned;
}

// tile_order_kernel costs ~8 us even when it only finds the frame uniform (one workgroup, a global round trip, the launch itself) —
// 1.3 % of a C2 step.  Frames of a size that keeps coming out uniform therefore run it only every 8th frame (the blend kernels use
// the XCD-contiguous order when no map was written); frames with uneven lists, where the map pays, run it always.  The verdicts
// reach the host through a pinned word the kernel writes (no copy, no wait; a stale value only delays the policy by a frame).
void run_tile_order(CapEntry* ce, const uint2* ranges, int gx, int gy, int* map, uint32_t* map_flag, int opt, hipStream_t s) {
    if (opt == 1) return;                  // XCD-contiguous order: the flag word is zero
    uint32_t* hv = pinned_u32();
    const int slot = ce ? (int)(ce - g_caps) : 0;
    uint32_t* verdict = nullptr;
    if (hv && opt == 0) {
        const uint32_t v = hv[R_SLOTS + slot];
        if (v == 1u) ce->uniform_streak = ce->uniform_streak < 1000 ? ce->uniform_streak + 1 : 1000;
        else if (v == 2u) ce->uniform_streak = 0;
        void* dv = nullptr;
        if (hipHostGetDevicePointer(&dv, hv + R_SLOTS + slot, 0) == hipSuccess) verdict = static_cast<uint32_t*>(dv);
        if (verdict && ce->uniform_streak >= 4 && (ce->frames & 7u) != 0u) return;
    }
    launch_tile_order(ranges, gx, gy, map, map_flag, opt, verdict, s);
}

int higher_msb(uint32_t n) {   // number of bits needed to represent values < n
    int b = 0;
    while ((1ull << b) < (unsigned long long)n) b++;
    return b < 1 ? 1 : b;
}

}  // namespace

namespace surfel {
int api_fail(int code, const char* what, hipError_t e) { return fail(code, what, e); }
}  // namespace surfel

extern "C" {

int surfel_abi_version(void) { return SURFEL_ABI_VERSION; }
const char* surfel_last_error(void) { return g_err.c_str(); }
const char* surfel_stage_name(int stage) { return (stage >= 0 && stage < 11) ? kStageNames[stage] : "?"; }
int surfel_last_stage_ms(float* ms, int cap) {
    int n = g_stage_n < cap ? g_stage_n : cap;
    for (int i = 0; i < n; i++) ms[i] = g_stage_ms[i];
    return n;
}
int surfel_last_stage_ids(int* ids, int cap) {
    int n = g_stage_n < cap ? g_stage_n : cap;
    for (int i = 0; i < n; i++) ids[i] = g_stage_id[i];
    return n;
}

int surfel_debug_sort_pairs(surfel_alloc_fn scratch_alloc, void* scratch_user, uint32_t* keys, uint32_t* vals, int64_t n,
                            int begin_bit, int end_bit, void* stream) {
    if (!scratch_alloc || n < 0 || (n > 0 && (!keys || !vals)) || begin_bit < 0 || end_bit > 32 || begin_bit > end_bit)
        return fail(SURFEL_E_INVALID, "bad arguments");
    if (n == 0) return 0;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const size_t sb = radix_sort_scratch_bytes((size_t)n);
    const size_t words = align_up((size_t)n * sizeof(uint32_t));
    char* base = static_cast<char*>(scratch_alloc(scratch_user, 2 * words + sb));
    if (!base) return fail(SURFEL_E_ALLOC, "sort scratch allocation failed");
    uint32_t* kb = reinterpret_cast<uint32_t*>(base);
    uint32_t* vb = reinterpret_cast<uint32_t*>(base + words);
    const int w = radix_sort_pairs_u32(keys, vals, kb, vb, (size_t)n, begin_bit, end_bit, base + 2 * words, s);
    if (w < 0) return fail(SURFEL_E_LIMIT, "sort size limit");
    if (w == 1) {
        HIP_TRY(hipMemcpyAsync(keys, kb, (size_t)n * sizeof(uint32_t), hipMemcpyDeviceToDevice, s));
        HIP_TRY(hipMemcpyAsync(vals, vb, (size_t)n * sizeof(uint32_t), hipMemcpyDeviceToDevice, s));
    }
    HIP_TRY(hipGetLastError());
    return 0;
}

int surfel_set_option(const char* name, int value) {
    if (name && std::strcmp(name, "cull") == 0) { g_opt_cull = value ? 1 : 0; return 0; }
    if (name && std::strcmp(name, "tile_depth_sort") == 0) { g_opt_tile_sort = value < 0 ? 0 : (value > 2 ? 2 : value); return 0; }
    if (name && std::strcmp(name, "large_sort") == 0) { set_large_sort_impl(value); return 0; }
    if (name && std::strcmp(name, "fwd_pipe") == 0) { set_fwd_pipe(value); return 0; }
    if (name && std::strcmp(name, "bwd_variant") == 0) { g_opt_bwd_variant = value < 0 ? 0 : (value > 3 ? 3 : value); return 0; }
    if (name && std::strcmp(name, "tile_stream") == 0) { g_opt_stream = value != 0; return 0; }
    if (name && std::strcmp(name, "capacity_binning") == 0) { g_opt_capacity = value != 0; return 0; }
    if (name && std::strcmp(name, "tile_order") == 0) { g_opt_tile_order = value < 0 ? 0 : (value > 2 ? 2 : value); return 0; }
    return fail(SURFEL_E_INVALID, "unknown option");
}

int surfel_set_backward_hook(surfel_hook_fn colour_ready, void* user) {
    g_colour_hook = colour_ready; g_colour_hook_user = user;
    return 0;
}

int surfel_debug_last_binning(void) { return g_last_binning; }
int surfel_debug_capacity_evictions(void) { return g_cap_evictions; }

int surfel_debug_image_layout(int width, int height, int64_t* out) {      // host arithmetic only: no device is touched
    if (width <= 0 || height <= 0 || !out) return fail(SURFEL_E_INVALID, "bad arguments");
    size_t total = 0;
    ImgState::carve(nullptr, width, height, &total);
    char* const base = reinterpret_cast<char*>(static_cast<uintptr_t>(1) << 40);      // (a carve over a fictitious base: only differences are used)
    const ImgState im = ImgState::carve(base, width, height, nullptr);
    out[0] = (int64_t)total;
    out[1] = reinterpret_cast<char*>(im.final_T) - base;
    out[2] = reinterpret_cast<char*>(im.n_contrib) - base;
    out[3] = reinterpret_cast<char*>(im.tile_map) - base;
    return 0;
}
int surfel_debug_set_blend_stats(void* dev_u64x8) { g_blend_stats = static_cast<unsigned long long*>(dev_u64x8); return 0; }

int surfel_collect_stage_ms(float* sum_ms, int* count, int cap) {
    for (int i = 0; i < cap; i++) { sum_ms[i] = 0.f; count[i] = 0; }
    std::lock_guard<std::mutex> lk(g_pending_mu);
    for (auto& p : g_pending) {
        float ms = 0.f;
        if (hipEventSynchronize(p.e1) == hipSuccess && hipEventElapsedTime(&ms, p.e0, p.e1) == hipSuccess && p.stage < cap) {
            sum_ms[p.stage] += ms; count[p.stage]++;
        }
        (void)hipEventDestroy(p.e0); (void)hipEventDestroy(p.e1);
    }
    g_pending.clear();
    return 11;
}

// Lazily counted frames (SURFEL_OPT_LAZY_COUNT): a capacity-path forward that did not wait for its instance count.  The count is
// collected by surfel_forward_count() — or by the next forward of this thread, which refuses to go on if that frame had overflowed
// its capacity without anybody looking (its images were built from truncated lists).
struct LazyPending { bool pending = false; int W = 0, H = 0, dev = -1; int64_t cap = 0; bool overflowed = false, host_total = false; int64_t R = 0; };
thread_local LazyPending g_lazy;

// waits for the pending count; returns the exact instance count (>= 0) and updates the per-size history, or a negative code
int64_t lazy_finish() {
    if (!g_lazy.pending) return g_lazy.R;
    // (the pair of the device the frame was rendered on — the thread may have made another device current since)
    PerDevice* pd = per_device(g_lazy.dev);
    uint32_t* hR = pd ? pd->pinned : nullptr;
    hipEvent_t evR = pd ? pd->ev : nullptr;
    if (!hR || !evR) return fail(SURFEL_E_HIP, "pinned buffer / event of the lazily counted frame's device not found");
    HIP_TRY(hipEventSynchronize(evR));
    int64_t R = 0;
    if (g_lazy.host_total) R = (int64_t)hR[kPinnedTotal];
    else for (int k = 0; k < R_SLOTS; k++) R += (int64_t)hR[k];
    g_lazy.pending = false;
    g_lazy.R = R;
    g_lazy.overflowed = R > g_lazy.cap;
    g_last_R = R; g_last_W = g_lazy.W; g_last_H = g_lazy.H;
    CapEntry* ce = cap_entry(g_lazy.W, g_lazy.H, true);
    if (ce->frames == ~0u) ce->frames = 0;
    if (g_lazy.overflowed) { if (ce->boost < 2) ce->boost++; ce->last_overflow = ce->frames; }
    ce->maxR = R > ce->maxR - ce->maxR / 64 ? R : ce->maxR - ce->maxR / 64;
    return R;
}

int64_t surfel_forward_count(void) {
    const int64_t R = lazy_finish();
    if (R < 0) return R;
    if (g_lazy.overflowed) {
        g_lazy.overflowed = false;      // reported: the caller redoes the frame
        return fail(SURFEL_E_OVERFLOW, "the lazily counted frame held more tile instances than its capacity: render it again (its outputs are incomplete)");
    }
    return R;
}

int64_t surfel_rasterize_forward(surfel_alloc_fn geom_alloc, void* geom_user, surfel_alloc_fn binning_alloc, void* binning_user,
                                 surfel_alloc_fn image_alloc, void* image_user, int P, int D, int M, const float* background,
                                 int width, int height, const float* means3D, const float* shs, const float* colors_precomp,
                                 const float* opacities, const float* scales, float scale_modifier, const float* rotations,
                                 const float* transMat_precomp, const float* viewmatrix, const float* projmatrix,
                                 const float* cam_pos, float tan_fovx, float tan_fovy, int prefiltered, float* out_color,
                                 float* out_others, int* radii, int debug, void* stream) {
    (void)tan_fovx; (void)tan_fovy; (void)prefiltered;
    g_stage_n = 0;
    if (g_lazy.pending) { const int64_t r = lazy_finish(); if (r < 0) return r; }
    if (g_lazy.overflowed) {
        g_lazy.overflowed = false;
        return fail(SURFEL_E_OVERFLOW, "the previous lazily counted frame overflowed its capacity and surfel_forward_count() was never called for it");
    }
    // per-call option overrides ride in the upper bits of `debug` (include/surfel_hip.h); the low byte is the debug mode
    const bool opt_lazy = (debug & SURFEL_OPT_LAZY_COUNT) != 0;
    const int opt_cull = (debug & SURFEL_OPT_NO_CULL) ? 0 : g_opt_cull;
    const int opt_tile_sort = ((debug >> 9) & 3) ? ((debug >> 9) & 3) - 1 : g_opt_tile_sort;
    const int opt_capacity = (debug & SURFEL_OPT_EXACT_BINNING) ? 0 : g_opt_capacity;
    const int opt_tile_order = ((debug >> 19) & 3) ? ((debug >> 19) & 3) - 1 : g_opt_tile_order;
    const bool opt_stream = !(debug & SURFEL_OPT_NO_STREAM);      // (the caller knows no backward follows: render.py-style inference, no_grad re-renders)
    const int map_len = tile_map_len((width + TILE - 1) / TILE, (height + TILE - 1) / TILE);
    debug &= 0xff;
    g_last_binning = 0;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (!geom_alloc || !binning_alloc || !image_alloc) return fail(SURFEL_E_INVALID, "allocator callback is NULL");
    if (P < 0 || width <= 0 || height <= 0) return fail(SURFEL_E_INVALID, "bad sizes");
    if (P > 0) {   // an empty scene (P == 0) renders the background; per-surfel pointers may then be NULL
        if ((shs == nullptr) == (colors_precomp == nullptr)) return fail(SURFEL_E_INVALID, "provide exactly one of shs / colors_precomp");
        const bool has_sr = scales != nullptr && rotations != nullptr;
        if (has_sr == (transMat_precomp != nullptr) || ((scales != nullptr) != (rotations != nullptr)))
            return fail(SURFEL_E_INVALID, "provide exactly one of (scales, rotations) / transMat_precomp");
        if (!means3D || !opacities || !viewmatrix || !projmatrix || !cam_pos || !radii) return fail(SURFEL_E_INVALID, "required pointer is NULL");
        if (D < 0 || D > 3 || (shs && M < (D + 1) * (D + 1))) return fail(SURFEL_E_INVALID, "bad SH degree / coefficient count");
    }
    if (!background || !out_color || !out_others) return fail(SURFEL_E_INVALID, "required pointer is NULL");
    const int gx = (width + TILE - 1) / TILE, gy = (height + TILE - 1) / TILE;
    if (gx > 1023 || gy > 1023) return fail(SURFEL_E_LIMIT, "image larger than 16368 px per side");
    const size_t HW = (size_t)width * height;

    size_t img_bytes = 0;
    ImgState::carve(nullptr, width, height, &img_bytes);
    void* img_base = image_alloc(image_user, img_bytes);
    if (!img_base) return fail(SURFEL_E_ALLOC, "image buffer allocation failed");
    ImgState img = ImgState::carve(img_base, width, height, nullptr);
    // (the carved span up to the next array: a multiple of 256 B, so the runtime needs ONE fill kernel, not an aligned part + a tail)
    HIP_TRY(hipMemsetAsync(img.ranges, 0, (size_t)(reinterpret_cast<char*>(img.final_T) - reinterpret_cast<char*>(img.ranges)), s));

    StageTimer tm(debug, s);
    int64_t R = 0;
    GeomState geom{};
    BinState bin{};
    CapEntry* tile_ce = cap_entry(width, height, true);      // per-size history (capacity, tile-order verdicts) of this host thread
    if (tile_ce->frames == ~0u) {      // a (re-)used slot must not inherit the previous size's tile-order verdict
        tile_ce->frames = 0;
        if (uint32_t* hv = pinned_u32()) hv[R_SLOTS + (int)(tile_ce - g_caps)] = 0u;
    }
    tile_ce->frames++;
    if (tile_ce->boost > 0 && tile_ce->frames - tile_ce->last_overflow > 256u) { tile_ce->boost--; tile_ce->last_overflow = tile_ce->frames; }
    if (P > 0) {
        // scratch sizes (host-side queries only): [depth-sort scratch | scan state]
        const size_t psort_bytes = align_up(radix_sort_scratch_bytes((size_t)P));
        const size_t temp_bytes = psort_bytes + scan_scratch_words((size_t)P) * sizeof(uint32_t);
        size_t geom_bytes = 0;
        GeomState::carve(nullptr, P, temp_bytes, &geom_bytes);
        void* geom_base = geom_alloc(geom_user, geom_bytes);
        if (!geom_base) return fail(SURFEL_E_ALLOC, "geometry buffer allocation failed");
        geom = GeomState::carve(geom_base, P, temp_bytes, nullptr);

        PreprocessArgs pa{};
        pa.P = P; pa.D = D; pa.M = M; pa.W = width; pa.H = height; pa.gx = gx; pa.gy = gy; pa.scale_modifier = scale_modifier;
        pa.cull = opt_cull;
        pa.pack_tiles = P < (1 << PACK_ID_BITS) ? 1 : 0;
        pa.means3D = means3D; pa.opacities = opacities; pa.scales = scales; pa.rotations = rotations;
        pa.transMat_precomp = transMat_precomp; pa.colors_precomp = colors_precomp; pa.shs = shs;
        pa.viewmatrix = viewmatrix; pa.projmatrix = projmatrix; pa.campos = cam_pos;
        pa.rec = geom.rec; pa.depths = geom.depths; pa.depth_keys = geom.dkey_a; pa.ident = geom.ord_a; pa.radii = radii;
        pa.tiles_touched = geom.tiles_touched; pa.clamped = geom.clamped; pa.total_instances = img.total; pa.rects = geom.rects;
        pa.shjac = (opt_stream && shs != nullptr && M == 16) ? geom.shjac : nullptr;
        uint32_t* scan_state = reinterpret_cast<uint32_t*>(geom.temp + psort_bytes);
        pa.zero_a = reinterpret_cast<uint32_t*>(geom.temp); pa.zero_a_words = (uint32_t)radix_sort_head_words((size_t)P);
        pa.zero_b = scan_state; pa.zero_b_words = (uint32_t)scan_scratch_words((size_t)P);

        // Binning plan, from history only (nothing of this frame is known on the host yet).
        //  * path: per-tile depth sort (small / medium frames: emit in surfel-index order, order every tile's run by depth afterwards
        //    in LDS — no P-sized radix sort) or depth-presorted emission (large frames: the original two-level scheme).  Both give the
        //    same per-tile order (depth bits, then surfel index); the choice is a speed heuristic on the previous frame's instances
        //    per tile (unknown on the first call: decided once R has arrived).
        //  * sizes: CAPACITY binning (per-tile path, <= 2^20 instances) sizes the binning buffers from the largest count recent frames
        //    of this size produced and never waits for this frame's count in the middle of the forward (surfel_sort.hip); otherwise
        //    the buffers are sized exactly, after a host wait for R.
        const int64_t ntiles_all = (int64_t)gx * gy;
        constexpr int64_t kTileSortMaxAvg = 640;      // bitonic work grows as n log^2 n: at ~1100 instances per tile it costs 0.25 ms vs 0.15 ms for the P-sized radix sort (C4)
        const int end_bit = higher_msb((uint32_t)(gx * gy));
        int per_tile = opt_tile_sort == 2 ? 1 : (opt_tile_sort == 0 ? 0 : -1);
        if (per_tile < 0 && g_last_R >= 0 && g_last_W == width && g_last_H == height) per_tile = g_last_R <= kTileSortMaxAvg * ntiles_all ? 1 : 0;
        CapEntry* ce = cap_entry(width, height, true);
        int64_t cap = 0;
        if (opt_capacity && per_tile == 1 && ce->maxR >= 0 && debug != 1) {
            cap = ce->maxR + (ce->maxR >> (3 - ce->boost)) + 4096;
            cap = (cap + 16383) / 16384 * 16384;      // stable sizes for the caller's caching allocator
            if (cap > ((int64_t)1 << 20) || !capacity_binning_ok((size_t)cap, end_bit)) cap = 0;
        }
        if (cap > 0) {      // the binning buffers exist before preprocess runs: it clears the tile sort's head on the way
            const size_t sort_bytes = capacity_sort_scratch_bytes((size_t)cap, end_bit);
            size_t bin_bytes = 0;
            const bool strm = opt_stream && BinState::wants_stream((size_t)cap);
            BinState::carve(nullptr, (size_t)cap, sort_bytes, &bin_bytes, strm);
            void* bin_base = binning_alloc(binning_user, bin_bytes);
            if (!bin_base) return fail(SURFEL_E_ALLOC, "binning buffer allocation failed");
            bin = BinState::carve(bin_base, (size_t)cap, sort_bytes, nullptr, strm);
            pa.zero_c = reinterpret_cast<uint32_t*>(bin.sort_temp); pa.zero_c_words = (uint32_t)bin_emit_head_words();
            pa.block_totals = geom.offsets;      // (the scan output of the exact path: free here)
        }
        tm.begin();
        launch_preprocess_fwd(pa, s);
        STAGE_END(tm, ST_PRE);
        // The instance count is copied to the host right behind preprocess.  Exact path: waited for after the depth sort + scan have
        // been enqueued.  Capacity path: looked at when the whole forward is enqueued (validation only).
        uint32_t* hR = pinned_u32();
        hipEvent_t evR = r_event();
        if (!hR || !evR) return fail(SURFEL_E_HIP, "pinned buffer / event creation failed");
        const bool host_total = cap > 0;      // capacity path: bin_emit_kernel stores the total into the mapped pinned buffer itself — no copy kernel
        if (!host_total) {
            HIP_TRY(hipMemcpyAsync(hR, img.total, sizeof(uint32_t) * R_SLOTS, hipMemcpyDeviceToHost, s));
            HIP_TRY(hipEventRecord(evR, s));
        }

        bool blended = false;
        if (cap > 0) {
            const uint32_t* n_dev = img.total + 2 * R_SLOTS;      // written by bin_emit_kernel
            const bool odd = radix_sort_result_buffer((size_t)cap, 0, end_bit) == 1;
            uint32_t* va = odd ? bin.vals_alt : bin.point_list;
            uint32_t* vb = odd ? bin.point_list : bin.vals_alt;
            tm.begin();
            launch_bin_emit(P, geom.tiles_touched, geom.rects, geom.offsets, geom.rec, bin.keys_a, va, gx, (size_t)cap, bin.sort_temp, end_bit,
                            img.total + 2 * R_SLOTS, host_total ? hR + kPinnedTotal : nullptr, s);
            if (host_total) HIP_TRY(hipEventRecord(evR, s));
            STAGE_END(tm, ST_EMIT);
            tm.begin();
            // the tile ranges come out of the sort's last pass (its scatter knows every key boundary); frames of one tile have no pass
            const bool fused_ranges = end_bit > 0;
            const int wk = radix_sort_pairs_u32_devn(bin.keys_a, va, bin.keys_b, vb, (size_t)cap, end_bit, n_dev, bin.sort_temp, fused_ranges ? img.ranges : nullptr, s);
            STAGE_END(tm, ST_SORT);
            if (!fused_ranges) {
                tm.begin();
                launch_tile_ranges_devn((size_t)cap, n_dev, wk ? bin.keys_b : bin.keys_a, img.ranges, s);
                STAGE_END(tm, ST_RANGES);
            }
            tm.begin();
            launch_tile_depth_sort(gx * gy, ce->maxR, img.ranges, bin.point_list, geom.dkey_a, bin.vals_alt, bin.keys_a, bin.keys_b, fused_ranges, s);
            STAGE_END(tm, ST_TSORT);
            run_tile_order(tile_ce, img.ranges, gx, gy, img.tile_map, img.total + 2 * R_SLOTS + 1, opt_tile_order, s);
            BlendFwdArgs ba{};
            ba.W = width; ba.H = height; ba.gx = gx; ba.gy = gy;
            ba.ranges = img.ranges; ba.point_list = bin.point_list; ba.rec = geom.rec; ba.bg = background;
            ba.out_color = out_color; ba.out_others = out_others; ba.final_T = img.final_T; ba.n_contrib = img.n_contrib;
            ba.tile_map = img.tile_map; ba.map_flag = img.total + 2 * R_SLOTS + 1; ba.map_len = map_len;
            ba.stats = g_blend_stats;
            ba.strm_rec = bin.strm_rec; ba.strm_mask = bin.strm_mask; ba.totals = img.total; ba.walk_word = img.total + 2 * R_SLOTS + 2;
            stream_register(bin.point_list, img_base, launch_blend_fwd_writes_stream(ba) ? bin.strm_rec : nullptr, bin.strm_mask, opt_stream && P > 0 && shs != nullptr && M == 16);
            tm.begin();
            launch_blend_fwd(ba, s);
            STAGE_END(tm, ST_BLEND);
            if (opt_lazy) {      // the count stays on its way: surfel_forward_count() (or the next forward) collects it
                g_lazy.pending = true; g_lazy.W = width; g_lazy.H = height; g_lazy.cap = cap; g_lazy.overflowed = false; g_lazy.host_total = host_total;
                if (hipGetDevice(&g_lazy.dev) != hipSuccess) g_lazy.dev = -1;
                g_last_binning = 4;
                HIP_TRY(hipGetLastError());
                return cap;
            }
            HIP_TRY(hipEventSynchronize(evR));      // the emission finished long ago; the device still holds the rest of the forward
            if (host_total) R = (int64_t)hR[kPinnedTotal];
            else for (int k = 0; k < R_SLOTS; k++) R += (int64_t)hR[k];
            g_last_binning = 1;
            blended = R <= cap;
            if (!blended) {      // overflow: the frame is redone with exact sizes below (same results as if it had taken that path at once)
                g_last_binning = 2;
                if (ce->boost < 2) ce->boost++;
                ce->last_overflow = ce->frames;
                HIP_TRY(hipMemsetAsync(img.ranges, 0, sizeof(uint2) * (size_t)gx * gy, s));
                // (the capacity path's device total is not this frame's count any more — the exact path leaves 0 there, so that
                // the backward's overflow guard compares nothing stale; the tile-map flag goes with it, run_tile_order decides again below)
                HIP_TRY(hipMemsetAsync(img.total + 2 * R_SLOTS, 0, 2 * sizeof(uint32_t), s));
            }
        }
        if (!blended) {
            const bool r_known = cap > 0;
            if (per_tile < 0 && !r_known) {         // first frame of this size: wait for R now (loses the host/device overlap once)
                HIP_TRY(hipEventSynchronize(evR));
                int64_t r0 = 0;
                for (int k = 0; k < R_SLOTS; k++) r0 += (int64_t)hR[k];
                per_tile = r0 <= kTileSortMaxAvg * ntiles_all ? 1 : 0;
            }
            tm.begin();
            const uint32_t* order = geom.ord_a;      // identity (written by preprocess)
            if (!per_tile) {
                // (1) surfel order by view depth (stable; culled surfels carry key 0xffffffff and sort last)
                const int which = radix_sort_pairs_u32(geom.dkey_a, geom.ord_a, geom.dkey_b, geom.ord_b, (size_t)P, 0, 32, geom.temp, s, true);
                if (which < 0) return fail(SURFEL_E_LIMIT, "too many surfels for the depth sort");
                order = which ? geom.ord_b : geom.ord_a;
            }
            // (2) instance offsets in emission order: inclusive scan of tiles_touched[order[k]]
            launch_scan_gather(geom.tiles_touched, order, pa.pack_tiles, geom.offsets, (size_t)P, scan_state, s);
            STAGE_END(tm, ST_SCAN);
            if (!r_known) {
                HIP_TRY(hipEventSynchronize(evR));
                for (int k = 0; k < R_SLOTS; k++) R += (int64_t)hR[k];
            }

            const size_t sort_bytes = radix_sort_scratch_bytes((size_t)R);
            size_t bin_bytes = 0;
            const bool strm = opt_stream && BinState::wants_stream((size_t)R);
            BinState::carve(nullptr, (size_t)R, sort_bytes, &bin_bytes, strm);
            void* bin_base = binning_alloc(binning_user, bin_bytes > 0 ? bin_bytes : 256);
            if (!bin_base) return fail(SURFEL_E_ALLOC, "binning buffer allocation failed");
            bin = BinState::carve(bin_base, (size_t)R, sort_bytes, nullptr, strm);
            if (R > 0) {
                // (3) stable sort on the tile-id bits only: depth order inside every tile is preserved.  The value buffers
                // are assigned so that the ping-pong ends in bin.point_list.
                const bool odd = radix_sort_result_buffer((size_t)R, 0, end_bit) == 1;      // result lands in the b buffers
                uint32_t* va = odd ? bin.vals_alt : bin.point_list;
                uint32_t* vb = odd ? bin.point_list : bin.vals_alt;
                tm.begin();
                launch_emit_instances(P, geom.rec, geom.rects, order, pa.pack_tiles ? ((1u << PACK_ID_BITS) - 1u) : 0xffffffffu, geom.offsets, bin.keys_a, va, gx, reinterpret_cast<uint32_t*>(bin.sort_temp),
                                      (uint32_t)radix_sort_head_words((size_t)R), s);
                STAGE_END(tm, ST_EMIT);
                tm.begin();
                const int wk = radix_sort_pairs_u32(bin.keys_a, va, bin.keys_b, vb, (size_t)R, 0, end_bit, bin.sort_temp, s, true);
                if (wk < 0) return fail(SURFEL_E_LIMIT, "too many tile instances for the tile sort");
                const uint32_t* sorted_keys = wk ? bin.keys_b : bin.keys_a;
                STAGE_END(tm, ST_SORT);
                tm.begin();
                launch_tile_ranges(R, sorted_keys, img.ranges, s);
                STAGE_END(tm, ST_RANGES);
                if (per_tile) {
                    // (4) every tile orders its run by (depth bits, surfel index); the ping-pong buffers of the tile sort are free now
                    tm.begin();
                    launch_tile_depth_sort(gx * gy, R, img.ranges, bin.point_list, geom.dkey_a, bin.vals_alt, bin.keys_a, bin.keys_b, false, s);
                    STAGE_END(tm, ST_TSORT);
                }
            }
        }
        g_last_R = R; g_last_W = width; g_last_H = height;
        g_lazy.R = R;
        ce->maxR = R > ce->maxR - ce->maxR / 64 ? R : ce->maxR - ce->maxR / 64;      // the largest recent count, slowly forgotten
        if (blended) { HIP_TRY(hipGetLastError()); return R; }
    } else {
        (void)geom_alloc(geom_user, 256);
        void* bin_base = binning_alloc(binning_user, 256);
        if (!bin_base) return fail(SURFEL_E_ALLOC, "binning buffer allocation failed");
        bin = BinState::carve(bin_base, 0, 0, nullptr);
        g_lazy.R = 0;      // (surfel_forward_count of an empty scene)
    }

    run_tile_order(tile_ce, img.ranges, gx, gy, img.tile_map, img.total + 2 * R_SLOTS + 1, opt_tile_order, s);
    BlendFwdArgs ba{};
    ba.W = width; ba.H = height; ba.gx = gx; ba.gy = gy;
    ba.ranges = img.ranges; ba.point_list = bin.point_list; ba.rec = geom.rec; ba.bg = background;
    ba.out_color = out_color; ba.out_others = out_others; ba.final_T = img.final_T; ba.n_contrib = img.n_contrib;
    ba.tile_map = img.tile_map; ba.map_flag = img.total + 2 * R_SLOTS + 1; ba.map_len = map_len;
    ba.stats = g_blend_stats;
    ba.avg_list = (int)(R / ((int64_t)gx * gy));
    ba.strm_rec = bin.strm_rec; ba.strm_mask = bin.strm_mask; ba.totals = img.total; ba.walk_word = img.total + 2 * R_SLOTS + 2;
    stream_register(bin.point_list, img_base, launch_blend_fwd_writes_stream(ba) ? bin.strm_rec : nullptr, bin.strm_mask, opt_stream && P > 0 && shs != nullptr && M == 16);
    tm.begin();
    launch_blend_fwd(ba, s);
    STAGE_END(tm, ST_BLEND);
    HIP_TRY(hipGetLastError());
    (void)HW;
    return R;
}

int surfel_rasterize_backward(surfel_alloc_fn scratch_alloc, void* scratch_user, int P, int D, int M, int64_t R,
                              const float* background, int width, int height, const float* means3D, const float* shs,
                              const float* colors_precomp, const float* scales, float scale_modifier, const float* rotations,
                              const float* transMat_precomp, const float* viewmatrix, const float* projmatrix,
                              const float* cam_pos, float tan_fovx, float tan_fovy, const int* radii, const void* geom_buffer,
                              const void* binning_buffer, const void* image_buffer, const float* dL_dout_color,
                              const float* dL_dout_others, float* dL_dmeans2D, float* dL_dnormal, float* dL_dopacity,
                              float* dL_dcolors, float* dL_dmeans3D, float* dL_dtransMat, float* dL_dsh, float* dL_dscales,
                              float* dL_drots, int debug, void* stream) {
    (void)tan_fovx; (void)tan_fovy; (void)colors_precomp;
    g_stage_n = 0;
    const int debug_in = debug;
    const int opt_variant = (debug & SURFEL_OPT_BWD_SCAN) ? 3 : ((debug & SURFEL_OPT_BWD_QUAD) ? 1 : ((debug & SURFEL_OPT_BWD_ROWS) ? 0 : g_opt_bwd_variant));   // 0 rows, 1 quad, 2 auto, 3 scan
    debug &= 0xff;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (P == 0) return 0;
    if (!scratch_alloc || !geom_buffer || !binning_buffer || !image_buffer) return fail(SURFEL_E_INVALID, "buffer / allocator is NULL");
    if (!dL_dout_color || !dL_dout_others || !dL_dmeans2D || !dL_dopacity || !dL_dcolors || !dL_dmeans3D)
        return fail(SURFEL_E_INVALID, "gradient pointer is NULL");
    if (transMat_precomp && !dL_dtransMat) return fail(SURFEL_E_INVALID, "dL_dtransMat is NULL but transMat_precomp was given");
    if (!transMat_precomp && (!dL_dscales || !dL_drots || !scales || !rotations)) return fail(SURFEL_E_INVALID, "scale/rotation pointers are NULL");
    const int gx = (width + TILE - 1) / TILE, gy = (height + TILE - 1) / TILE;

    GeomState geom = GeomState::carve(const_cast<void*>(geom_buffer), P, 0, nullptr);   // rocPRIM scratch is last: layout of the rest is size-independent
    BinState bin = BinState::carve(const_cast<void*>(binning_buffer), (size_t)R, 0, nullptr);
    ImgState img = ImgState::carve(const_cast<void*>(image_buffer), width, height, nullptr);

    // per-instance gradient records: blend_bwd writes the record of every instance up to its tile's cut exactly once (no memset, no
    // atomics); the cuts (8 B per tile) sit behind the records
    const size_t grec_bytes = align_up((size_t)(R > 0 ? R : 1) * GREC_F * sizeof(float));
    const size_t cut_bytes = align_up((size_t)gx * gy * sizeof(uint2));
    float* grec = static_cast<float*>(scratch_alloc(scratch_user, grec_bytes + cut_bytes + (size_t)P));
    if (!grec) return fail(SURFEL_E_ALLOC, "gradient record allocation failed");
    // tile cuts instead of zero records: only where the records that are never written are worth a test in front of every fetch
    // (measured: +17 us on preprocess_bwd at 0.5 M instances / 1.7 per surfel, -4 % on blend_bwd and preprocess_bwd at 8 M / 4 per surfel)
    const bool use_cut = (debug_in & SURFEL_OPT_TILE_CUTS) ? true : ((debug_in & SURFEL_OPT_ZERO_RECORDS) ? false : R >= ((int64_t)1 << 21));
    uint2* cut = use_cut ? reinterpret_cast<uint2*>(reinterpret_cast<char*>(grec) + grec_bytes) : nullptr;
    // ... and one byte per surfel "has a record at all" (crowded frames stage a few per cent of their instances: most surfels none)
    uint8_t* has_rec = use_cut ? reinterpret_cast<uint8_t*>(grec) + grec_bytes + cut_bytes : nullptr;
    if (has_rec) HIP_TRY(hipMemsetAsync(has_rec, 0, (size_t)P, s));
    StageTimer tm(debug, s);

    BlendBwdArgs bb{};
    bb.W = width; bb.H = height; bb.gx = gx; bb.gy = gy;
    bb.ranges = img.ranges; bb.point_list = bin.point_list; bb.rec = geom.rec; bb.bg = background;
    bb.final_T = img.final_T; bb.n_contrib = img.n_contrib; bb.dL_dpix = dL_dout_color; bb.dL_dothers = dL_dout_others;
    bb.tile_map = img.tile_map; bb.map_flag = img.total + 2 * R_SLOTS + 1; bb.map_len = tile_map_len(gx, gy);      // the forward's tile order (its lists are the backward's lists)
    bb.grec = grec; bb.cut = cut; bb.has_rec = has_rec; bb.depths = geom.depths; bb.variant = opt_variant; bb.stats = g_blend_stats; bb.walk_word = img.total + 2 * R_SLOTS + 2;
    // num_rendered of a lazily counted frame is its CAPACITY: if the frame's real total (on the device since bin_emit_kernel; 0 on the
    // exact path) exceeds it, the lists are truncated and the first-instance slots run past `grec` — the kernels below return at once
    bool have_jac = false;
    {
        const float4* srec = nullptr; const uint32_t* smask = nullptr;
        const bool found = stream_lookup(binning_buffer, image_buffer, &srec, &smask, &have_jac);
        if (found && !(debug_in & SURFEL_OPT_BWD_GATHER) && g_opt_stream) { bb.strm_rec = srec; bb.strm_mask = smask; }      // (else NULL: the walks gather)
    }
    bb.n_dev = img.total + 2 * R_SLOTS; bb.n_cap = (uint32_t)(R < 0xffffffffll ? R : 0xffffffffll);
    if (R > 0) {
        // auto: rows or scan.  The scan walk takes the frames whose footprints span many tiles and the large ones — decided ON THE DEVICE
        // from the frame's totals (surfel_blend_bwd.h: device_picks_scan; the host of a lazily counted frame knows neither R nor the
        // emitting surfels): both kernels are launched and the workgroups of one of them return at once (~4 us).  A rule, not a timed
        // choice: the walks differ in summation order, and which bits a frame gets must follow from the frame alone.
        // Where the host knows the count for certain — exact binning; a lazily counted frame reports its capacity, <= 2^20 — the size half
        // of the rule is applied here and only ONE kernel is launched (an idle grid of a 4K frame's 32 k workgroups costs ~0.1 ms):
        // 2^21 <= R < 2^26 -> the scan walk alone, R >= 2^26 -> rows alone (C5: 1.3e8 instances of which 4 % are staged — scan 3.19 vs
        // 3.13 ms); below 2^21 both, and the device decides by footprint.  (The per-quad walk, bit-identical to rows, is launched only
        // on request: it never won on a frame the rule would give it — profiles/r04_bench_n1_full.json blend_bwd_ms_by_walk.)
        const bool auto_walk = opt_variant == 2;
        bb.variant = auto_walk ? 0 : opt_variant;
        if (auto_walk && !g_blend_stats) {
            if (R >= ((int64_t)1 << 21)) bb.variant = R < ((int64_t)1 << 26) ? 3 : 0;
            else bb.scan_rule = 1;
        }
        tm.begin(ST_BBWD);
        launch_blend_bwd(bb, s);
        STAGE_END(tm, ST_BBWD);
    }

    PreprocessBwdArgs pb{};
    pb.P = P; pb.D = D; pb.M = M; pb.W = width; pb.H = height; pb.scale_modifier = scale_modifier;
    // record gather: per thread, or by the wave when a surfel holds many records AND the records (80 B each) overflow the 256 MB
    // Infinity Cache (measured: the cooperative form wins at C5 only, and loses 30 - 85 % on small frames); bit-identical sums
    pb.coop = (debug_in & SURFEL_OPT_PBWD_COOP) ? 1 : ((debug_in & SURFEL_OPT_PBWD_THREAD) ? 0 : ((R >= (int64_t)6 * P && R >= ((int64_t)32 << 20)) ? 1 : 0));
    pb.shjac = (have_jac && !(debug_in & SURFEL_OPT_PBWD_NO_JAC)) ? geom.shjac : nullptr;
    pb.means3D = means3D; pb.radii = radii; pb.shs = shs; pb.clamped = geom.clamped; pb.scales = scales; pb.rotations = rotations;
    pb.transMat_precomp = transMat_precomp; pb.viewmatrix = viewmatrix; pb.projmatrix = projmatrix; pb.campos = cam_pos;
    pb.rec = geom.rec; pb.tiles_touched = geom.tiles_touched; pb.grec = grec; pb.cut = cut; pb.has_rec = has_rec; pb.depths = geom.depths; pb.gx = gx;
    pb.n_dev = bb.n_dev; pb.n_cap = bb.n_cap;
    pb.dL_dtransMat = dL_dtransMat; pb.dL_dnormal = dL_dnormal; pb.dL_dopacity = dL_dopacity; pb.dL_dcolors = dL_dcolors;
    pb.dL_dsh = dL_dsh; pb.dL_dmeans2D = dL_dmeans2D; pb.dL_dmeans3D = dL_dmeans3D; pb.dL_dscales = dL_dscales; pb.dL_drots = dL_drots;
    tm.begin();
    if (g_colour_hook) {      // dL/dcolour first, so the caller can start moving it while the geometry chain rule runs
        launch_colour_gradients(pb, s);
        pb.keep_colors = 1;
        g_colour_hook(g_colour_hook_user);
    }
    launch_preprocess_bwd(pb, s);
    STAGE_END(tm, ST_PBWD);
    HIP_TRY(hipGetLastError());
    return 0;
}

int surfel_mark_visible(int P, const float* means3D, const float* viewmatrix, const float* projmatrix, uint8_t* present,
                        void* stream) {
    (void)projmatrix;
    if (P < 0 || (P > 0 && (!means3D || !viewmatrix || !present))) return fail(SURFEL_E_INVALID, "bad arguments");
    launch_mark_visible(P, means3D, viewmatrix, present, static_cast<hipStream_t>(stream));
    HIP_TRY(hipGetLastError());
    return 0;
}

int surfel_knn_dist2(surfel_alloc_fn scratch_alloc, void* scratch_user, int P, const float* points, float* mean_dist2,
                     void* stream) {
    if (P < 0 || (P > 0 && (!points || !mean_dist2 || !scratch_alloc))) return fail(SURFEL_E_INVALID, "bad arguments");
    if (P == 0) return 0;
    const size_t bytes = knn_scratch_bytes(P);
    void* scratch = scratch_alloc(scratch_user, bytes);
    if (!scratch) return fail(SURFEL_E_ALLOC, "knn scratch allocation failed");
    launch_knn(P, points, mean_dist2, scratch, bytes, static_cast<hipStream_t>(stream));
    HIP_TRY(hipGetLastError());
    return 0;
}

}  // extern "C"
