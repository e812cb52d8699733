"""Drop-in replacement for the `diff_surfel_rasterization` Python package of hbb1/2d-gaussian-splatting.

The reference imports these two names literally (/root/reference/gaussian_renderer/__init__.py:14) and
uses them at :37-53 and :97-106; the original package is an absent submodule
(/root/reference/.gitmodules:1-3).  Same names, argument meaning and error behaviour; the native half
is libsurfel_hip.so (hand-written HIP for gfx950) reached through the C ABI in include/surfel_hip.h.
No CPU / PyTorch fallback exists: importing this without the built library raises ImportError.
"""
from typing import NamedTuple

import torch
import torch.nn as nn

import surfel_native as _n

_n.load()  # fail loudly at import time if the HIP extension is missing

last_num_rendered = 0   # instance count R of the most recent forward (introspection for bench / tests)
_last_image = None      # (image buffer, tiles x, tiles y) of the most recent forward


def tile_row_instances():
    """[tile rows] int64 device tensor: tile instances per 16-row tile row of the most recent forward (read from its tile
    ranges) — the weights surfel_dist.band_bounds balances the row bands with.  None before the first forward."""
    if _last_image is None:
        return None
    buf, gx, gy = _last_image
    r = buf[:gx * gy * 8].view(torch.int32).view(gy, gx, 2).to(torch.int64)
    return (r[..., 1] - r[..., 0]).sum(1)


IMG_HEAD_U2 = 64 + 2      # uint2 slots behind the tile ranges in the image buffer's zeroed head (csrc/surfel_api.hip: ImgState::carve):
                          # 2 x 64 partial counters, the capacity path's instance total + the tile-map flag, the frame's backward walk + a spare word


def image_layout(width, height):
    """White-box byte offsets into the image buffer: (gx, gy, final_T, n_contrib, tile_map) — for tests and statistics only."""
    gx, gy = (width + 15) // 16, (height + 15) // 16
    al = lambda v: (v + 255) // 256 * 256
    final_T = al((gx * gy + IMG_HEAD_U2) * 8)
    n_contrib = al(final_T + 12 * width * height)
    tile_map = al(n_contrib + 8 * width * height)
    return gx, gy, final_T, n_contrib, tile_map


def staged_instances(width, height):
    """Σ over tiles of the deepest list position any pixel of the tile composited (max n_contrib) of the most recent forward: the
    instances a blend pass has to read — the rest of a tile's list lies behind its saturation point.  White-box read of the image
    buffer (csrc/surfel_api.hip: ImgState = ranges + 2 x 64 partial counters + 1 | final_T 3HW | n_contrib 2HW, 256-B aligned)."""
    if _last_image is None:
        return None
    buf, gx, gy = _last_image
    off = image_layout(width, height)[3]
    last = buf[off:off + 4 * width * height].view(torch.int32).view(height, width)
    pad = torch.zeros((gy * 16, gx * 16), dtype=torch.int32, device=buf.device)
    pad[:height, :width] = last
    return int(pad.view(gy, 16, gx, 16).amax(dim=(1, 3)).sum().item())


_grad_arena = None


def set_grad_arena(arena):
    """Optional (multi-GPU): a dict of preallocated fp32 tensors — any of means3D [P,3], sh [P,M,3], opacities [P,1],
    scales [P,2], rotations [P,4], colors [P,3] (dL/dcolour; in SH mode the clamp-masked dL/d(SH colour) that
    surfel_sh_grad_gather exchanges), means2D [P,3] (the densification statistic) — that the backward writes its gradients into and returns, instead of fresh tensors.
    surfel_dist.GradBucket.arena() hands out views of ONE flat buffer, so the gradient all-reduce needs no packing pass.
    The kernels write every element, so the tensors need no zeroing.  An explicit `sh=None` entry makes the backward skip the
    SH-coefficient gradients altogether (their autograd gradient is then None).  An optional `_owner` tensor scopes the arena to
    one model: it is used only by backwards whose means3D shares that tensor's storage (GaussianModel.bind passes its parameter
    store), every other caller gets fresh gradient tensors.  An arena tensor that is used but does not fit the backward (shape, dtype,
    device, contiguity) raises: a stale binding must not silently drop gradients.  None restores the default."""
    global _grad_arena
    _grad_arena = arena


class GaussianRasterizationSettings(NamedTuple):
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: torch.Tensor
    scale_modifier: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    sh_degree: int
    campos: torch.Tensor
    prefiltered: bool
    debug: bool


def _c(t):
    """contiguous fp32 view (or None for the reference's 'empty tensor' placeholders)."""
    if t is None or t.numel() == 0:
        return None
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


def finish_count():
    """Collect the instance count of a lazily counted forward (debug bit surfel_native.OPT_LAZY_COUNT; include/surfel_hip.h:
    surfel_forward_count) on the calling thread: returns it, or raises surfel_native.CapacityOverflow — the frame is then rendered
    again with OPT_EXACT_BINNING and whatever was derived from it recomputed."""
    global last_num_rendered
    last_num_rendered = _n.forward_count()
    return last_num_rendered


def rasterize_gaussians(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, raster_settings):
    # no backward can follow (torch.no_grad(), or nothing requires a gradient: the reference's render.py / training_report renders,
    # /root/reference/render.py:57, train.py:211): the forward leaves no tile stream behind (include/surfel_hip.h: SURFEL_OPT_NO_STREAM)
    if not (torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in
                                            (means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp))):
        raster_settings = raster_settings._replace(debug=int(raster_settings.debug) | _n.OPT_NO_STREAM)
    return _RasterizeGaussians.apply(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                                     raster_settings)


class _RasterizeGaussians(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, raster_settings):
        rs = raster_settings
        lib = _n.load()
        dev = means3D.device
        if dev.type != "cuda":
            raise RuntimeError("diff_surfel_rasterization: tensors must live on a HIP device (got %s)" % dev)
        means3D = _c(means3D); sh = _c(sh); colors_precomp = _c(colors_precomp); opacities = _c(opacities)
        scales = _c(scales); rotations = _c(rotations); cov3Ds_precomp = _c(cov3Ds_precomp)
        bg = _c(rs.bg); viewmatrix = _c(rs.viewmatrix); projmatrix = _c(rs.projmatrix); campos = _c(rs.campos)
        P = 0 if means3D is None else means3D.shape[0]
        H, W = int(rs.image_height), int(rs.image_width)
        M = 0 if sh is None else sh.shape[1]
        out_color = torch.empty((3, H, W), dtype=torch.float32, device=dev)
        out_others = torch.empty((7, H, W), dtype=torch.float32, device=dev)
        radii = torch.empty((P,), dtype=torch.int32, device=dev)
        ga, ba, ia = _n.TorchAllocator(dev), _n.TorchAllocator(dev), _n.TorchAllocator(dev)
        with torch.cuda.device(dev):
            R = lib.surfel_rasterize_forward(ga.cb, None, ba.cb, None, ia.cb, None, P, int(rs.sh_degree), M, _n.ptr(bg), W, H,
                                             _n.ptr(means3D), _n.ptr(sh), _n.ptr(colors_precomp), _n.ptr(opacities),
                                             _n.ptr(scales), float(rs.scale_modifier), _n.ptr(rotations),
                                             _n.ptr(cov3Ds_precomp), _n.ptr(viewmatrix), _n.ptr(projmatrix), _n.ptr(campos),
                                             float(rs.tanfovx), float(rs.tanfovy), int(bool(rs.prefiltered)),
                                             _n.ptr(out_color), _n.ptr(out_others), _n.ptr(radii), int(rs.debug),
                                             _n.current_stream_ptr(dev))
        if R < 0:
            raise RuntimeError("surfel_rasterize_forward failed (%d): %s" % (R, _n.last_error()))
        global last_num_rendered, _last_image
        last_num_rendered = int(R)
        _last_image = (ia.last(), (W + 15) // 16, (H + 15) // 16)
        ctx.raster_settings = rs
        ctx.num_rendered = int(R)
        ctx.dims = (P, M, H, W)
        ctx.has = (sh is not None, colors_precomp is not None, scales is not None, cov3Ds_precomp is not None)
        geomBuffer, binningBuffer, imgBuffer = ga.last(), ba.last(), ia.last()
        none = torch.empty(0, device=dev)
        ctx.save_for_backward(*(x if x is not None else none for x in
                                (colors_precomp, means3D, scales, rotations, cov3Ds_precomp, radii, sh, geomBuffer,
                                 binningBuffer, imgBuffer, bg, viewmatrix, projmatrix, campos)))
        ctx.mark_non_differentiable(radii)
        ctx.set_materialize_grads(False)      # unused outputs arrive as None in backward instead of zero-filled tensors
        return out_color, radii, out_others

    @staticmethod
    def backward(ctx, grad_out_color, grad_radii, grad_depth):
        rs = ctx.raster_settings
        lib = _n.load()
        (colors_precomp, means3D, scales, rotations, cov3Ds_precomp, radii, sh, geomBuffer, binningBuffer, imgBuffer, bg,
         viewmatrix, projmatrix, campos) = ctx.saved_tensors
        has_sh, has_col, has_sr, has_cov = ctx.has
        P, M, H, W = ctx.dims
        dev = means3D.device
        z = lambda *shape: torch.empty(shape, dtype=torch.float32, device=dev)   # the kernels write every element
        arena = _grad_arena or {}
        owner = arena.get("_owner")
        if owner is not None and owner.untyped_storage().data_ptr() != means3D.untyped_storage().data_ptr():
            arena = {}      # this backward belongs to another model / caller than the one that bound the arena: plain fresh tensors

        def out(name, *shape):
            t = arena.get(name)
            if t is None:
                return z(*shape)          # no arena tensor for this output: the default behaviour
            if tuple(t.shape) != shape or t.dtype != torch.float32 or t.device != dev or not t.is_contiguous():
                # the arena belongs to THIS model (the owner check above passed) but does not fit: a stale binding (the model was
                # re-sized without bind()) — a fresh tensor here would be dropped by the parameter gate and training would silently
                # receive no gradient
                raise RuntimeError("diff_surfel_rasterization: grad arena tensor %r is %s %s on %s, the backward needs contiguous float32 %s on %s "
                                   "(re-bind the arena after resizing the model)" % (name, tuple(t.shape), t.dtype, t.device, shape, dev))
            return t
        g_means2D, g_normal, g_colors = out("means2D", P, 3), None, out("colors", P, 3)      # dL/dnormal: internal, not requested
        g_opac = out("opacities", P, 1)
        g_means3D = out("means3D", P, 3)
        g_trans = z(P, 9) if has_cov else None      # an intermediate unless cov3D_precomp carries the gradient
        skip_sh = has_sh and "sh" in arena and arena["sh"] is None     # caller rebuilds dL/dSH from dL/dcolour (include/surfel_train.h)
        g_sh = out("sh", P, M, 3) if (has_sh and not skip_sh) else None
        g_scales = out("scales", P, 2) if has_sr else None
        g_rots = out("rotations", P, 4) if has_sr else None
        gc = grad_out_color.contiguous().float() if grad_out_color is not None else torch.zeros((3, H, W), device=dev)
        gd = grad_depth.contiguous().float() if grad_depth is not None else torch.zeros((7, H, W), device=dev)
        sa = _n.TorchAllocator(dev)
        opt = lambda t, ok: _n.ptr(t) if ok else None
        with torch.cuda.device(dev):
            rc = lib.surfel_rasterize_backward(sa.cb, None, P, int(rs.sh_degree), M, ctx.num_rendered, _n.ptr(bg), W, H,
                                               _n.ptr(means3D), opt(sh, has_sh), opt(colors_precomp, has_col),
                                               opt(scales, has_sr), float(rs.scale_modifier), opt(rotations, has_sr),
                                               opt(cov3Ds_precomp, has_cov), _n.ptr(viewmatrix), _n.ptr(projmatrix),
                                               _n.ptr(campos), float(rs.tanfovx), float(rs.tanfovy), _n.ptr(radii),
                                               _n.ptr(geomBuffer), _n.ptr(binningBuffer), _n.ptr(imgBuffer), _n.ptr(gc),
                                               _n.ptr(gd), _n.ptr(g_means2D), _n.ptr(g_normal), _n.ptr(g_opac),
                                               _n.ptr(g_colors), _n.ptr(g_means3D), _n.ptr(g_trans), _n.ptr(g_sh),
                                               _n.ptr(g_scales), _n.ptr(g_rots), int(rs.debug),
                                               _n.current_stream_ptr(dev))
        if rc < 0:
            raise RuntimeError("surfel_rasterize_backward failed (%d): %s" % (rc, _n.last_error()))
        return (g_means3D, g_means2D, g_sh, g_colors if has_col else None, g_opac, g_scales, g_rots,
                g_trans if has_cov else None, None)


class GaussianRasterizer(nn.Module):
    def __init__(self, raster_settings):
        super().__init__()
        self.raster_settings = raster_settings

    def markVisible(self, positions):
        rs = self.raster_settings
        with torch.no_grad():
            positions = _c(positions)
            P = positions.shape[0]
            present = torch.zeros((P,), dtype=torch.uint8, device=positions.device)
            with torch.cuda.device(positions.device):
                rc = _n.load().surfel_mark_visible(P, _n.ptr(positions), _n.ptr(_c(rs.viewmatrix)), _n.ptr(_c(rs.projmatrix)),
                                                   _n.ptr(present), _n.current_stream_ptr(positions.device))
            if rc < 0:
                raise RuntimeError("surfel_mark_visible failed: %s" % _n.last_error())
        return present.bool()

    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                cov3D_precomp=None):
        rs = self.raster_settings
        if (shs is None and colors_precomp is None) or (shs is not None and colors_precomp is not None):
            raise Exception('Please provide excatly one of either SHs or precomputed colors!')
        if ((scales is None or rotations is None) and cov3D_precomp is None) or \
                ((scales is not None or rotations is not None) and cov3D_precomp is not None):
            raise Exception('Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!')
        return rasterize_gaussians(means3D, means2D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp, rs)
