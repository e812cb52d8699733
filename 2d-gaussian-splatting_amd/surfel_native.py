"""ctypes loader for libsurfel_hip.so — the C-ABI boundary declared in include/surfel_hip.h.

There is NO fallback: if the HIP library is missing or cannot be loaded this module raises, so a
GPU box can never silently run some other path.  torch is imported first so that the HIP runtime
already mapped by PyTorch-ROCm (same SONAME libamdhip64.so.7) is the one our library binds to; torch
here is plumbing only (device memory through its caching allocator, current stream).
"""
import ctypes as C
import os
import threading

import torch  # noqa: F401  (must precede the CDLL below, see module doc)

_HERE = os.path.dirname(os.path.abspath(__file__))
# SURFEL_LIB: diagnostics only (e.g. the -DSURFEL_IEEE_MATH twin built by `python build.py --ieee`)
LIB_PATH = os.environ.get("SURFEL_LIB") or os.path.join(_HERE, "lib", "libsurfel_hip.so")

ALLOC_FN = C.CFUNCTYPE(C.c_void_p, C.c_void_p, C.c_size_t)
HOOK_FN = C.CFUNCTYPE(None, C.c_void_p)

EXPORTS = ["surfel_abi_version", "surfel_last_error", "surfel_rasterize_forward", "surfel_rasterize_backward",
           "surfel_mark_visible", "surfel_knn_dist2", "surfel_last_stage_ms", "surfel_last_stage_ids", "surfel_stage_name",
           "surfel_collect_stage_ms", "surfel_set_option", "surfel_debug_sort_pairs", "surfel_debug_set_blend_stats", "surfel_debug_last_binning", "surfel_debug_capacity_evictions", "surfel_debug_image_layout", "surfel_debug_box_probe", "surfel_debug_latency_probe", "surfel_set_backward_hook", "surfel_forward_count",
           # include/surfel_train.h
           "surfel_l1_ssim_forward", "surfel_l1_ssim_backward", "surfel_l1_ssim_forward_w", "surfel_l1_ssim_backward_w", "surfel_render_post_forward", "surfel_render_post_backward", "surfel_train_loss_forward", "surfel_train_loss_backward",
           "surfel_reduce_partials", "surfel_loss_finalize", "surfel_activate", "surfel_adam_step", "surfel_train_update", "surfel_sh_grad_gather", "surfel_densify_stats"]

# per-call option overrides carried in the upper bits of the `debug` argument (include/surfel_hip.h)
OPT_NO_CULL = 1 << 8
OPT_BWD_QUAD = 1 << 11
OPT_BWD_ROWS = 1 << 12
OPT_PBWD_COOP = 1 << 13
OPT_PBWD_THREAD = 1 << 14
OPT_EXACT_BINNING = 1 << 16    # forward: size the binning buffers exactly (host wait for the instance count) for this call
OPT_TILE_CUTS = 1 << 17        # backward: tile cuts instead of zero gradient records (default: R >= 2^21); bit-identical
OPT_ZERO_RECORDS = 1 << 18     # backward: zero gradient records behind a tile's saturation point (default: R < 2^21)
OPT_LAZY_COUNT = 1 << 21       # forward: do not wait for the instance count; forward_count() collects it (include/surfel_hip.h)
E_OVERFLOW = -5
OPT_BWD_GATHER = 1 << 22       # backward: ignore the forward's tile stream, gather by surfel id (bit-identical)
OPT_NO_STREAM = 1 << 23        # forward: no backward follows (inference / no_grad): leave no tile stream behind
OPT_PBWD_NO_JAC = 1 << 24      # backward: read the SH block again instead of the forward's d(colour)/d(direction) rows (same result to rounding)
OPT_BWD_SCAN = 1 << 15         # scan walk (lanes = instances); deterministic, not bit-identical to rows / quad


def opt_tile_order(mode):
    """per-call "tile_order" (0 auto, 1 XCD-contiguous, 2 longest lists first) in the debug word (SURFEL_OPT_TILE_ORDER)"""
    return ((int(mode) + 1) & 3) << 19


def opt_tile_sort(mode):
    return ((mode + 1) & 3) << 9


_lib = None
_lock = threading.Lock()


def load():
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "libsurfel_hip.so not found at %s — build it with `python 2d-gaussian-splatting_amd/build.py` "
                "(hipcc --offload-arch=gfx950). There is no CPU / PyTorch fallback for the rasterizer." % LIB_PATH)
        lib = C.CDLL(LIB_PATH)
        vp, f, i, i64 = C.c_void_p, C.c_float, C.c_int, C.c_int64
        lib.surfel_abi_version.restype = i
        lib.surfel_last_error.restype = C.c_char_p
        lib.surfel_rasterize_forward.restype = i64
        lib.surfel_rasterize_forward.argtypes = [ALLOC_FN, vp, ALLOC_FN, vp, ALLOC_FN, vp, i, i, i, vp, i, i,
                                                 vp, vp, vp, vp, vp, f, vp, vp, vp, vp, vp, f, f, i, vp, vp, vp, i, vp]
        lib.surfel_rasterize_backward.restype = i
        lib.surfel_rasterize_backward.argtypes = [ALLOC_FN, vp, i, i, i, i64, vp, i, i, vp, vp, vp, vp, f, vp, vp, vp, vp, vp,
                                                  f, f, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i, vp]
        lib.surfel_mark_visible.restype = i
        lib.surfel_mark_visible.argtypes = [i, vp, vp, vp, vp, vp]
        lib.surfel_knn_dist2.restype = i
        lib.surfel_knn_dist2.argtypes = [ALLOC_FN, vp, i, vp, vp, vp]
        lib.surfel_last_stage_ms.restype = i
        lib.surfel_last_stage_ms.argtypes = [C.POINTER(C.c_float), i]
        lib.surfel_last_stage_ids.restype = i
        lib.surfel_last_stage_ids.argtypes = [C.POINTER(C.c_int), i]
        lib.surfel_collect_stage_ms.restype = i
        lib.surfel_collect_stage_ms.argtypes = [C.POINTER(C.c_float), C.POINTER(C.c_int), i]
        lib.surfel_stage_name.restype = C.c_char_p
        lib.surfel_stage_name.argtypes = [i]
        lib.surfel_debug_sort_pairs.restype = i
        lib.surfel_debug_sort_pairs.argtypes = [ALLOC_FN, vp, vp, vp, i64, i, i, vp]
        lib.surfel_debug_set_blend_stats.restype = i
        lib.surfel_debug_set_blend_stats.argtypes = [vp]
        lib.surfel_debug_box_probe.restype = i
        lib.surfel_debug_box_probe.argtypes = [vp, i64, C.POINTER(C.c_float), vp]
        lib.surfel_debug_latency_probe.restype = i
        lib.surfel_debug_latency_probe.argtypes = [vp, i64, i, C.POINTER(C.c_float), vp]
        lib.surfel_set_backward_hook.restype = i
        lib.surfel_set_backward_hook.argtypes = [HOOK_FN, vp]
        lib.surfel_forward_count.restype = i64
        lib.surfel_forward_count.argtypes = []
        lib.surfel_set_option.restype = i
        lib.surfel_set_option.argtypes = [C.c_char_p, i]
        # ---- include/surfel_train.h
        fp = C.POINTER(C.c_float)
        for name, args in (("surfel_l1_ssim_forward", [i, i, i, vp, vp, vp, vp, vp]),
                           ("surfel_l1_ssim_backward", [i, i, i, vp, vp, vp, f, f, vp, vp, vp, vp]),
                           ("surfel_l1_ssim_forward_w", [i, i, i, i, vp, vp, vp, vp, vp]),
                           ("surfel_l1_ssim_backward_w", [i, i, i, i, vp, vp, vp, f, f, vp, vp, vp, vp]),
                           ("surfel_render_post_forward", [i, i, vp, vp, f, vp, vp, vp]),
                           ("surfel_render_post_backward", [i, i, vp, vp, f, vp, f, f, vp, vp, vp]),
                           ("surfel_train_loss_forward", [i, i, vp, vp, vp, vp, vp, vp, f, vp, vp]),
                           ("surfel_train_loss_backward", [i, i, vp, vp, vp, f, f, vp, vp, f, f, f, vp, vp, vp, vp, vp, f, f, f, vp, vp, vp]),
                           ("surfel_reduce_partials", [vp, i, i, i, f, vp, vp]),
                           ("surfel_loss_finalize", [vp, i, i, vp, i, i, f, f, f, vp, vp, vp]),
                           ("surfel_activate", [i, vp, vp, vp]),
                           ("surfel_adam_step", [i, vp, vp, vp, vp, vp, fp, f, f, f, i, f, i, i, vp, vp, i, vp]),
                           ("surfel_train_update", [i, vp, vp, vp, vp, vp, fp, f, f, f, i, f, i, i, vp, vp, vp, vp, vp, vp, vp, vp]),
                           ("surfel_sh_grad_gather", [i, i, i, vp, vp, vp, vp, vp]),
                           ("surfel_densify_stats", [i, vp, vp, vp, vp, vp, vp])):
            fn = getattr(lib, name)
            fn.restype = i
            fn.argtypes = args
        if lib.surfel_abi_version() != 1:
            raise ImportError("libsurfel_hip.so ABI version mismatch")
        # SURFEL_OPTIONS="name=value,..." -> surfel_set_option (A/B runs of an unmodified caller, e.g. bench.py)
        for kv in filter(None, os.environ.get("SURFEL_OPTIONS", "").split(",")):
            name, _, value = kv.partition("=")
            if lib.surfel_set_option(name.strip().encode(), int(value)) != 0:
                raise ValueError("SURFEL_OPTIONS: unknown option %r" % name)
        _lib = lib
    return _lib


_hook_keepalive = None


def set_backward_hook(fn):
    """surfel_set_backward_hook: fn() is called inside every rasterizer backward once dL/dcolour is final on the stream (None
    removes the hook).  The ctypes thunk is kept alive here for as long as the hook is installed."""
    global _hook_keepalive
    lib = load()
    if fn is None:
        lib.surfel_set_backward_hook(HOOK_FN(), None)
        _hook_keepalive = None
        return
    thunk = HOOK_FN(lambda user: fn())
    lib.surfel_set_backward_hook(thunk, None)
    _hook_keepalive = thunk


class CapacityOverflow(RuntimeError):
    """a lazily counted frame held more tile instances than its capacity: render it again (OPT_EXACT_BINNING)"""


def forward_count():
    """surfel_forward_count: exact instance count of this thread's last forward; raises CapacityOverflow for a lazily counted frame
    whose lists were truncated."""
    r = load().surfel_forward_count()
    if r == E_OVERFLOW:
        raise CapacityOverflow(last_error())
    if r < 0:
        raise RuntimeError("surfel_forward_count failed: %s" % last_error())
    return int(r)


def last_error():
    return load().surfel_last_error().decode()


def bucket_bytes(n):
    """Size actually requested from torch for an n-byte scratch buffer.  The R-sized buffers (binning state, the 80 B/instance
    gradient records: ~10 GB at 1.3e8 instances) change by a few per cent from frame to frame; the caching allocator cannot grow
    a cached block, so every new maximum cost a fresh hipMalloc of the whole buffer (290 ms at 10 GB, measured) while the slightly
    smaller block stayed cached — reserved memory ratcheted up by the buffer size each time.  Sizes of 64 MiB and more are
    therefore rounded up to the next multiple of 1/16 .. 1/8 of themselves (a power of two), so a fluctuating size settles in one or
    two blocks; smaller requests are left to the allocator's own 2 MiB rounding."""
    n = int(n)
    if n < (1 << 26):
        return max(n, 1)
    step = 1 << (n.bit_length() - 4)
    return (n + step - 1) // step * step


class TorchAllocator:
    """Allocator callback backed by torch's caching allocator (uint8 tensors kept alive in `held`).
    The ctypes callback closes over the `held` list only — not over `self` — so there is no reference cycle and the
    buffers go back to the caching allocator as soon as the last reference dies (with a cycle they waited for the
    cyclic GC, which at 10 M surfels grew the footprint by ~13 GB per step)."""

    def __init__(self, device):
        held = []

        def _alloc(user, nbytes):
            try:
                t = torch.empty(bucket_bytes(nbytes), dtype=torch.uint8, device=device)
            except Exception:  # out of memory -> NULL -> SURFEL_E_ALLOC
                return None
            held.append(t)
            return t.data_ptr()

        self.device = device
        self.held = held
        self.cb = ALLOC_FN(_alloc)

    def last(self):
        return self.held[-1]


def ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def current_stream_ptr(device):
    # (torch.cuda.current_stream() builds a Stream object per call: ~7 us of the ~400 us a training iteration costs on the host)
    if _raw_stream is not None and device.index is not None:
        return C.c_void_p(_raw_stream(device.index))
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


class ManualCtx:
    """Stand-in for the autograd context when a caller drives forward() / backward() of the library's autograd Functions itself
    (surfel_trainer: the iteration's chain is fixed — rasterizer -> loss -> loss backward -> rasterizer backward — and the autograd
    engine, its worker-thread hand-over and the parameter gates cost more host time than all the launches together)."""
    manual = True
    needs_input_grad = ()

    def save_for_backward(self, *tensors):
        self.saved_tensors = tensors

    def mark_non_differentiable(self, *tensors):
        pass

    def set_materialize_grads(self, value):
        pass


def stage_times():
    """[(stage_name, ms)] of the last debug-mode forward/backward on this thread."""
    lib = load()
    ms = (C.c_float * 16)()
    ids = (C.c_int * 16)()
    n = lib.surfel_last_stage_ms(ms, 16)
    lib.surfel_last_stage_ids(ids, 16)
    return [(lib.surfel_stage_name(ids[k]).decode(), float(ms[k])) for k in range(n)]


def collect_stage_times():
    """{stage_name: (total_ms, launches)} for all debug==2 calls since the previous collect (synchronises)."""
    lib = load()
    ms = (C.c_float * 16)()
    cnt = (C.c_int * 16)()
    n = lib.surfel_collect_stage_ms(ms, cnt, 16)
    return {lib.surfel_stage_name(k).decode(): (float(ms[k]), int(cnt[k])) for k in range(n) if cnt[k] > 0}
