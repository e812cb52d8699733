"""GaussianModel on MI355X — same public surface as the reference's scene/gaussian_model.py (getters :95-118, create_from_pcd
:124-146, training_setup :148-166, update_learning_rate :168-174, save/load_ply :193-255, reset_opacity :209-212,
densify_and_prune :389-403, add_densification_stats :405-407, capture/restore :58-92) but laid out for the GPU:

  * ONE flat fp32 store of 58 floats/surfel, planar by section  xyz 3P | opacity P | scaling 2P | rotation 4P | sh 48P
    (include/surfel_train.h).  Raw parameters, gradients, Adam moments and the multi-GPU all-reduce bucket share this
    layout, so an optimiser step is ONE fused HIP launch pair (surfel_adam_step: activation backward + Adam + next
    iteration's activations) instead of 6 parameter groups x ~10 kernels, and the gradient all-reduce is ONE collective on
    the very buffer the rasterizer's backward wrote (zero copies).
  * `_features_dc` / `_features_rest` are views of the interleaved [P,16,3] SH block the rasterizer reads directly — the
    reference's `torch.cat((f_dc, f_rest), dim=1)` per iteration (192 B/surfel copied forward, again backward) is gone.
  * activations (exp / normalize / sigmoid) are produced by the optimiser kernel for the NEXT iteration; their backward is
    folded into the same kernel.

Gradients never materialise as `.grad` attributes: they live in `self.grad` (same layout).  There is no CPU path for the
optimiser; the densification logic is plain torch indexing (runs every 100 iterations, off the hot path).
"""
import ctypes as C
import math
import os

import numpy as np
import torch

import surfel_native as _n
from simple_knn._C import distCUDA2

SECTIONS = (("xyz", 3), ("opacity", 1), ("scaling", 2), ("rotation", 4), ("sh", 48))
GEOM_FLOATS = 10     # xyz + opacity + scaling + rotation: the contiguous prefix that view-parallel training all-reduces
FLOATS = 58
SH_C0 = 0.28209479177387814          # utils/sh_utils.py:26
GROUPS = ("xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation")   # Adam groups, scene/gaussian_model.py:153-160


def inverse_sigmoid(x):
    return torch.log(x / (1 - x))


def expon_lr(step, lr_init, lr_final, lr_delay_steps=0, lr_delay_mult=1.0, max_steps=1000000):
    """Log-linear learning-rate decay with optional warm-up (the schedule of utils/general_utils.py:39-70)."""
    if step < 0 or (lr_init == 0.0 and lr_final == 0.0):
        return 0.0
    warm = 1.0
    if lr_delay_steps > 0:
        warm = lr_delay_mult + (1 - lr_delay_mult) * math.sin(0.5 * math.pi * min(max(step / lr_delay_steps, 0.0), 1.0))
    t = min(max(step / max_steps, 0.0), 1.0)
    return warm * math.exp(math.log(lr_init) * (1 - t) + math.log(lr_final) * t)


def quat_to_rotmat(q):
    """[K,4] (w,x,y,z), normalised here -> [K,3,3] (utils/general_utils.py:78-100)."""
    q = q / q.norm(dim=1, keepdim=True)
    w, x, y, z = q.unbind(1)
    return torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y),
                        2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
                        2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)], dim=1).reshape(-1, 3, 3)


class _ParamGate(torch.autograd.Function):
    """Marks store views as differentiable inputs of the rasterizer without creating `.grad` copies: the rasterizer's
    backward writes into the gradient store (set_grad_arena) and the tensors arriving here are dropped."""

    @staticmethod
    def forward(ctx, anchor, t):
        ctx.set_materialize_grads(False)      # a skipped gradient (None) must not be turned into a zero-filled tensor
        return t.view_as(t)

    @staticmethod
    def backward(ctx, g):
        return None, None


class _ParamSink(torch.autograd.Function):
    """Marks a store view as a differentiable input of ordinary PyTorch code (the compute_cov3D_python homography of
    gaussian_renderer/__init__.py:64-75) and ADDS the gradient autograd hands back into the gradient store's section — the
    counterpart of _ParamGate for gradients that do not come out of the rasterizer's own backward."""

    @staticmethod
    def forward(ctx, anchor, t, dst):
        ctx.dst = dst
        ctx.set_materialize_grads(False)
        return t.view_as(t)

    @staticmethod
    def backward(ctx, g):
        if g is not None:
            ctx.dst.add_(g.reshape(ctx.dst.shape))
        return None, None, None


def _views(buf, P):
    out, off = {}, 0
    for name, n in SECTIONS:
        out[name] = buf[off:off + P * n].view(P, n)
        off += P * n
    return out


COLOUR_FLOATS = 3    # the clamp-masked dL/dcolour rides right behind the geometry prefix (fused SH mode): 13 floats = 52 B / surfel


def exchange_same_view(grad, P, group=None, async_op=False):
    """Tile-band sharding: every rank rendered rows of the SAME view, so the SH gradient of the summed loss is
    basis(dir) (x) sum_r g_r — the colour gradients are simply ADDED.  With the colour block aliased behind the geometry prefix
    (GaussianModel.bind(sh_grad=False)) the whole per-surfel exchange is ONE all-reduce of 13 floats = 52 B / surfel."""
    import torch.distributed as dist
    return dist.all_reduce(grad[:(GEOM_FLOATS + COLOUR_FLOATS) * P], op=dist.ReduceOp.SUM, group=group, async_op=async_op)


def exchange_collectives(grad, gcol, P, group=None, async_op=False):
    """The two data-path collectives of a view-parallel step: all-gather of the per-rank colour gradients -> [world, P, 3] and
    all-reduce(SUM) of the geometry prefix of the flat gradient store (in place).  The gather is issued first: with
    async_op=True the caller gets (gall, gather_work, reduce_work) and can update the SH block as soon as the gather has landed
    while the all-reduce is still in flight."""
    import torch.distributed as dist
    world = dist.get_world_size(group)
    gall = torch.empty((world, P, 3), dtype=torch.float32, device=gcol.device)
    try:
        wg = dist.all_gather_into_tensor(gall, gcol.contiguous(), group=group, async_op=async_op)
    except (RuntimeError, NotImplementedError):          # backends without the flat form
        wg = dist.all_gather([gall[r] for r in range(world)], gcol.contiguous(), group=group, async_op=async_op)
    wr = dist.all_reduce(grad[:GEOM_FLOATS * P], op=dist.ReduceOp.SUM, group=group, async_op=async_op)
    return (gall, wg, wr) if async_op else gall


class GaussianModel:
    def __init__(self, sh_degree: int, device="cuda"):
        self.active_sh_degree = 0
        self.max_sh_degree = sh_degree
        if not 0 <= sh_degree <= 3:
            raise ValueError("sh_degree must be 0..3 (/root/reference/arguments/__init__.py:49)")
        # the store always holds 16 coefficients per surfel; with --sh_degree d < 3 only the first (d+1)^2 are ever active: the rest
        # stay zero (zero gradient -> zero Adam update) and are neither written to nor read from .ply / checkpoints
        self.n_coef = (sh_degree + 1) ** 2
        self.device = torch.device(device)
        self.P = 0
        self.theta = self.act = self.grad = self.m = self.v = None
        self.max_radii2D = torch.empty(0)
        self.xyz_gradient_accum = torch.empty(0)
        self.denom = torch.empty(0)
        self.percent_dense = 0
        self.spatial_lr_scale = 0
        self.step_count = 0                # optimiser steps taken (torch Adam's per-parameter `step`)
        self.lr = None                     # per-group learning rates, order GROUPS
        self.betas, self.eps = (0.9, 0.999), 1e-15
        self._anchor = None
        self._lr_args = None

    # ------------------------------------------------------------------ store
    def _alloc(self, P):
        dev = self.device
        self.P = P
        self.theta = torch.empty(P * FLOATS, dtype=torch.float32, device=dev)
        self.act = torch.empty(P * 7, dtype=torch.float32, device=dev)
        self._pv = _views(self.theta, P)
        self._av = dict(opacity=self.act[:P].view(P, 1), scaling=self.act[P:3 * P].view(P, 2), rotation=self.act[3 * P:].view(P, 4))
        self._anchor = torch.zeros((), device=dev, requires_grad=True)

    def _alloc_training(self):
        P, dev = self.P, self.device
        self.grad = torch.zeros(P * FLOATS, dtype=torch.float32, device=dev)
        self.m = torch.zeros(P * FLOATS, dtype=torch.float32, device=dev)
        self.v = torch.zeros(P * FLOATS, dtype=torch.float32, device=dev)
        self._gv = _views(self.grad, P)
        self._gcol_own = None
        self.gcol = self._gcol_alias()       # clamp-masked dL/dcolour of this rank's view

    def _gcol_alias(self):
        """[P,3] view of the first 3P floats of the gradient store's SH section: in fused-SH mode nothing else lives there (the SH
        gradients are rebuilt in registers), and geometry prefix + colour block form one contiguous 52 B/surfel exchange buffer."""
        P = self.P
        return self.grad[GEOM_FLOATS * P:(GEOM_FLOATS + COLOUR_FLOATS) * P].view(P, 3)

    def _set(self, xyz, f_dc, f_rest, opacity, scaling, rotation):
        P = xyz.shape[0]
        self._alloc(P)
        pv = self._pv
        pv["xyz"].copy_(xyz.reshape(P, 3))
        sh = pv["sh"].view(P, 16, 3)
        sh.zero_()
        sh[:, :1].copy_(f_dc.reshape(P, 1, 3))
        n_rest = f_rest.numel() // (3 * P) if P else 0
        if n_rest not in (self.n_coef - 1, 15):
            raise ValueError("f_rest holds %d coefficients per surfel, expected %d (sh_degree %d)" % (n_rest, self.n_coef - 1, self.max_sh_degree))
        sh[:, 1:1 + n_rest].copy_(f_rest.reshape(P, n_rest, 3))
        pv["opacity"].copy_(opacity.reshape(P, 1)); pv["scaling"].copy_(scaling.reshape(P, 2)); pv["rotation"].copy_(rotation.reshape(P, 4))
        self.refresh_activations()

    def refresh_activations(self):
        """act = sigmoid(opacity) | exp(scaling) | normalize(rotation) (scene/gaussian_model.py:95-115) — one HIP launch."""
        if self.device.type != "cuda":
            raise RuntimeError("GaussianModel: the parameter store must live on a HIP device (no CPU path)")
        with torch.cuda.device(self.device):
            rc = _n.load().surfel_activate(self.P, _n.ptr(self.theta), _n.ptr(self.act), _n.current_stream_ptr(self.device))
        if rc < 0:
            raise RuntimeError("surfel_activate failed: %s" % _n.last_error())

    # raw parameter views (the reference's nn.Parameters)
    @property
    def _xyz(self): return self._pv["xyz"]
    @property
    def _features_dc(self): return self._pv["sh"].view(self.P, 16, 3)[:, :1]
    @property
    def _features_rest(self): return self._pv["sh"].view(self.P, 16, 3)[:, 1:self.n_coef]
    @property
    def _opacity(self): return self._pv["opacity"]
    @property
    def _scaling(self): return self._pv["scaling"]
    @property
    def _rotation(self): return self._pv["rotation"]

    def _gate(self, t):
        if torch.is_grad_enabled() and self.grad is not None:
            return _ParamGate.apply(self._anchor, t)
        return t

    # what render() consumes (scene/gaussian_model.py:95-118)
    @property
    def get_xyz(self): return self._gate(self._pv["xyz"])
    @property
    def get_features(self): return self._gate(self._pv["sh"].view(self.P, 16, 3))
    @property
    def get_opacity(self): return self._gate(self._av["opacity"])
    @property
    def get_scaling(self): return self._gate(self._av["scaling"])
    @property
    def get_rotation(self): return self._gate(self._av["rotation"])

    def get_covariance(self, scaling_modifier=1):
        """splat2world [P,4,4] as the reference builds it (scene/gaussian_model.py:27-33), for compute_cov3D_python.
        While training (gradient store bound) the matrix is built from differentiable views: autograd carries the rasterizer's
        dL/dcov3D_precomp back through this PyTorch code and _ParamSink adds it into the store's xyz / scaling / rotation
        sections (w.r.t. the activated scaling / rotation, as the optimiser kernel expects) — the reference trains this path
        through autograd as well."""
        P = self.P
        scaling, rotation, xyz = self._av["scaling"], self._av["rotation"], self._pv["xyz"]
        if torch.is_grad_enabled() and self.grad is not None:
            gv = self._gv
            gv["scaling"].zero_(); gv["rotation"].zero_()        # the rasterizer's backward does not write them on this path
            scaling = _ParamSink.apply(self._anchor, scaling, gv["scaling"])
            rotation = _ParamSink.apply(self._anchor, rotation, gv["rotation"])
            xyz = _ParamSink.apply(self._anchor, xyz, gv["xyz"])  # added AFTER the rasterizer's backward has written its own share
        s = torch.cat([scaling * scaling_modifier, torch.ones_like(scaling)], dim=-1)[:, :3]
        RS = (quat_to_rotmat(rotation) * s[:, None, :]).permute(0, 2, 1)
        top = torch.cat([RS, torch.zeros((P, 3, 1), dtype=torch.float32, device=self.device)], dim=2)
        bottom = torch.cat([xyz, torch.ones((P, 1), dtype=torch.float32, device=self.device)], dim=1)[:, None, :]
        return torch.cat([top, bottom], dim=1)

    def oneupSHdegree(self):
        if self.active_sh_degree < self.max_sh_degree:
            self.active_sh_degree += 1

    # ------------------------------------------------------------------ initialisation
    def create_from_pcd(self, pcd, spatial_lr_scale: float):
        """pcd: object with .points [P,3] and .colors [P,3] in [0,1] (utils/graphics_utils.BasicPointCloud).
        scene/gaussian_model.py:124-146: DC colour -> SH, scale = sqrt(mean 3-NN dist^2) (HIP knn), random rotation,
        opacity 0.1."""
        self.spatial_lr_scale = spatial_lr_scale
        dev = self.device
        pts = torch.as_tensor(np.asarray(pcd.points)).float().to(dev)
        col = (torch.as_tensor(np.asarray(pcd.colors)).float().to(dev) - 0.5) / SH_C0
        P = pts.shape[0]
        dist2 = torch.clamp_min(distCUDA2(pts), 0.0000001)
        scales = torch.log(torch.sqrt(dist2))[..., None].repeat(1, 2)
        rots = torch.rand((P, 4), device=dev)
        opac = inverse_sigmoid(0.1 * torch.ones((P, 1), dtype=torch.float32, device=dev))
        self._set(pts, col.reshape(P, 1, 3), torch.zeros((P, self.n_coef - 1, 3), device=dev), opac, scales, rots)
        self.max_radii2D = torch.zeros((P,), device=dev)

    def set_parameters(self, xyz, f_dc, f_rest, opacity, scaling, rotation):
        """Raw (pre-activation) parameters from tensors / arrays (e.g. a loaded checkpoint)."""
        t = lambda a: torch.as_tensor(np.asarray(a) if not torch.is_tensor(a) else a).float().to(self.device)
        self._set(t(xyz), t(f_dc), t(f_rest), t(opacity), t(scaling), t(rotation))
        self.max_radii2D = torch.zeros((self.P,), device=self.device)

    # ------------------------------------------------------------------ optimiser
    def training_setup(self, training_args):
        """scene/gaussian_model.py:148-166: statistics buffers, six learning rates, Adam(eps=1e-15), xyz schedule."""
        a = training_args
        self.percent_dense = a.percent_dense
        self.xyz_gradient_accum = torch.zeros((self.P, 1), device=self.device)
        self.denom = torch.zeros((self.P, 1), device=self.device)
        self.lr = [a.position_lr_init * self.spatial_lr_scale, a.feature_lr, a.feature_lr / 20.0, a.opacity_lr, a.scaling_lr, a.rotation_lr]
        self._lr_args = dict(lr_init=a.position_lr_init * self.spatial_lr_scale, lr_final=a.position_lr_final * self.spatial_lr_scale,
                             lr_delay_mult=a.position_lr_delay_mult, max_steps=a.position_lr_max_steps)
        self._alloc_training()
        self.bind()

    def bind(self, sh_grad=True):
        """Point the rasterizer's backward at this model's gradient store (zero-copy).  sh_grad=False: the backward skips the
        192 B/surfel SH gradients; optimizer_step(colour_grads=...) rebuilds them in registers from the colour gradients."""
        import diff_surfel_rasterization as dsr
        gv = self._gv
        if sh_grad:       # explicit SH gradients occupy the SH section: the colour gradients need a tensor of their own
            if self._gcol_own is None or self._gcol_own.shape[0] != self.P:
                self._gcol_own = torch.zeros((self.P, 3), dtype=torch.float32, device=self.device)
            self.gcol = self._gcol_own
        else:
            self.gcol = self._gcol_alias()
        dsr.set_grad_arena(dict(means3D=gv["xyz"], sh=gv["sh"].view(self.P, 16, 3) if sh_grad else None, opacities=gv["opacity"],
                                scales=gv["scaling"], rotations=gv["rotation"], colors=self.gcol, _owner=self.theta))

    def update_learning_rate(self, iteration):
        lr = expon_lr(iteration, **self._lr_args)
        self.lr[0] = lr
        return lr

    def optimizer_step(self, grad_scale=1.0, colour_grads=None, parts=3):
        """optimizer.step() + zero_grad (train.py:136-138) as fused launches; refreshes the activations.
        colour_grads = (campos_all [N,3], gcol_all [N,P,3]): the SH block's gradients are rebuilt from the views' colour
        gradients inside the kernel (N = 1: this rank's view; N > 1: after exchange_collectives) instead of read from self.grad."""
        if parts & 1:               # parts: 1 = SH block, 2 = geometry sections; a 1-then-2 pair is one step
            self.step_count += 1
        lr = (C.c_float * 6)(*self.lr)
        cam = gc = None
        N = 0
        if colour_grads is not None:
            cam, gc = colour_grads[0].contiguous().float(), colour_grads[1].contiguous().float()
            N = int(gc.shape[0])
        with torch.cuda.device(self.device):
            rc = _n.load().surfel_adam_step(self.P, _n.ptr(self.theta), _n.ptr(self.grad), _n.ptr(self.m), _n.ptr(self.v), _n.ptr(self.act),
                                            lr, self.betas[0], self.betas[1], self.eps, self.step_count, float(grad_scale),
                                            int(self.active_sh_degree), N, _n.ptr(cam), _n.ptr(gc), int(parts), _n.current_stream_ptr(self.device))
        if rc < 0:
            raise RuntimeError("surfel_adam_step failed: %s" % _n.last_error())

    def update_step(self, colour_grads, stats=None, grad_scale=1.0):
        """add_densification_stats + optimizer_step(parts=3) as ONE launch (surfel_train_update) for the iterations in which nothing is
        rebuilt between the two; same bits.  colour_grads as in optimizer_step (required: the SH block is rebuilt in the kernel);
        stats = (dL/dmeans2D [P,3], radii [P] int32) or None."""
        self.step_count += 1
        lr = (C.c_float * 6)(*self.lr)
        cam, gc = colour_grads[0].contiguous().float(), colour_grads[1].contiguous().float()
        g = r = None
        if stats is not None:
            g, r = stats[0].contiguous().float(), stats[1].contiguous().to(torch.int32)
        with torch.cuda.device(self.device):
            rc = _n.load().surfel_train_update(self.P, _n.ptr(self.theta), _n.ptr(self.grad), _n.ptr(self.m), _n.ptr(self.v), _n.ptr(self.act),
                                               lr, self.betas[0], self.betas[1], self.eps, self.step_count, float(grad_scale),
                                               int(self.active_sh_degree), int(gc.shape[0]), _n.ptr(cam), _n.ptr(gc), _n.ptr(g), _n.ptr(r),
                                               _n.ptr(self.xyz_gradient_accum), _n.ptr(self.denom), _n.ptr(self.max_radii2D),
                                               _n.current_stream_ptr(self.device))
        if rc < 0:
            raise RuntimeError("surfel_train_update failed: %s" % _n.last_error())

    def exchange_gradients(self, campos_all, group=None):
        """View-parallel step: make self.grad the SUM over ranks of the per-view gradients with
          * ONE all-reduce of the contiguous geometry prefix (xyz, opacity, scaling, rotation: 40 B/surfel) and
          * ONE all-gather of the clamp-masked dL/dcolour (12 B/surfel/rank) from which every rank rebuilds the 48 SH gradients
            (surfel_sh_grad_gather) — exact, summed in rank order, so replicas stay bit-identical —
        instead of all-reducing all 232 B/surfel.  campos_all [world,3]: camera centres of the ranks' views of this step."""
        gall = exchange_collectives(self.grad, self.gcol, self.P, group)
        self.sh_grad_from_colours(campos_all, gall)

    def sh_grad_from_colours(self, campos_all, gcol_all):
        """grad.sh = sum_r basis(dir(xyz, campos_all[r])) (x) gcol_all[r]  for the active SH degree (one HIP launch)."""
        c = campos_all.contiguous().float(); g = gcol_all.contiguous().float()
        if g.untyped_storage().data_ptr() == self.grad.untyped_storage().data_ptr():
            g = g.clone()        # the fused-mode colour block lives inside the SH section this call overwrites
        with torch.cuda.device(self.device):
            rc = _n.load().surfel_sh_grad_gather(self.P, int(self.active_sh_degree), int(g.shape[0]), _n.ptr(self._pv["xyz"]), _n.ptr(c),
                                                 _n.ptr(g), _n.ptr(self._gv["sh"]), _n.current_stream_ptr(self.device))
        if rc < 0:
            raise RuntimeError("surfel_sh_grad_gather failed: %s" % _n.last_error())

    # ------------------------------------------------------------------ densification (scene/gaussian_model.py:257-407)
    def add_densification_stats(self, viewspace_point_tensor, update_filter=None, radii=None):
        """train.py:126-128 in one launch: max_radii2D, xyz_gradient_accum, denom for the surfels with radii > 0.
        `viewspace_point_tensor.grad` is the rasterizer's dL/dmeans2D statistic; `radii` the int32 radii of the view."""
        g = viewspace_point_tensor.grad if hasattr(viewspace_point_tensor, "grad") and viewspace_point_tensor.grad is not None \
            else viewspace_point_tensor
        if radii is None:
            raise ValueError("add_densification_stats needs the view's radii (visibility = radii > 0)")
        g = g.contiguous().float(); r = radii.contiguous().to(torch.int32)
        with torch.cuda.device(self.device):
            rc = _n.load().surfel_densify_stats(self.P, _n.ptr(g), _n.ptr(r), _n.ptr(self.xyz_gradient_accum), _n.ptr(self.denom),
                                                _n.ptr(self.max_radii2D), _n.current_stream_ptr(self.device))
        if rc < 0:
            raise RuntimeError("surfel_densify_stats failed: %s" % _n.last_error())

    def _rebuild(self, keep, new=None):
        """New store = rows `keep` (bool mask or None = all) of the old one, followed by the rows in `new`
        (dict section -> [K,n] raw values); Adam moments follow (zeros for new rows); statistics are NOT touched."""
        P = self.P
        idx = None if keep is None else torch.nonzero(keep.reshape(-1), as_tuple=False).reshape(-1)
        K = 0 if new is None else new["xyz"].shape[0]
        Pn = (P if idx is None else idx.numel()) + K
        old_p, old_m, old_v = self._pv, _views(self.m, P) if self.m is not None else None, _views(self.v, P) if self.v is not None else None
        theta = torch.empty(Pn * FLOATS, dtype=torch.float32, device=self.device)
        nv = _views(theta, Pn)
        m = v = None
        if self.m is not None:
            m = torch.zeros(Pn * FLOATS, dtype=torch.float32, device=self.device); v = torch.zeros_like(m)
            mv, vv = _views(m, Pn), _views(v, Pn)
        for name, n in SECTIONS:
            rows = old_p[name] if idx is None else old_p[name][idx]
            nk = rows.shape[0]
            nv[name][:nk].copy_(rows)
            if K:
                nv[name][nk:].copy_(new[name].reshape(K, n))
            if m is not None:
                mv[name][:nk].copy_(old_m[name] if idx is None else old_m[name][idx])
                vv[name][:nk].copy_(old_v[name] if idx is None else old_v[name][idx])
        self.P = Pn
        self.theta = theta
        self.act = torch.empty(Pn * 7, dtype=torch.float32, device=self.device)
        self._pv = nv
        self._av = dict(opacity=self.act[:Pn].view(Pn, 1), scaling=self.act[Pn:3 * Pn].view(Pn, 2), rotation=self.act[3 * Pn:].view(Pn, 4))
        if m is not None:
            self.m, self.v = m, v
            self.grad = torch.zeros(Pn * FLOATS, dtype=torch.float32, device=self.device)
            self._gv = _views(self.grad, Pn)
            self._gcol_own = None
            self.gcol = self._gcol_alias()
            self.bind()
        self._activate_host_or_device()
        return idx

    def _activate_host_or_device(self):
        self.refresh_activations()

    def prune_points(self, mask):
        """Remove the surfels where mask is True (scene/gaussian_model.py:290-305)."""
        keep = ~mask.reshape(-1)
        self._rebuild(keep)
        self.xyz_gradient_accum = self.xyz_gradient_accum[keep]
        self.denom = self.denom[keep]
        self.max_radii2D = self.max_radii2D[keep]

    def densification_postfix(self, new):
        """Append new surfels and reset the statistics of ALL surfels (scene/gaussian_model.py:329-346)."""
        self._rebuild(None, new)
        self.xyz_gradient_accum = torch.zeros((self.P, 1), device=self.device)
        self.denom = torch.zeros((self.P, 1), device=self.device)
        self.max_radii2D = torch.zeros((self.P,), device=self.device)

    def _rows(self, mask, repeat=1):
        return {name: self._pv[name][mask].repeat(repeat, 1) for name, _ in SECTIONS}

    def densify_and_clone(self, grads, grad_threshold, scene_extent):
        """Duplicate small surfels with a large view-space gradient (scene/gaussian_model.py:373-387)."""
        sel = (torch.norm(grads, dim=-1) >= grad_threshold) & (self._av["scaling"].max(dim=1).values <= self.percent_dense * scene_extent)
        self.densification_postfix(self._rows(sel))

    def densify_and_split(self, grads, grad_threshold, scene_extent, N=2, generator=None):
        """Replace large surfels with a large gradient by N samples inside them (scene/gaussian_model.py:348-371)."""
        n_init = self.P
        padded = torch.zeros((n_init,), device=self.device)
        padded[:grads.shape[0]] = grads.squeeze()
        sel = (padded >= grad_threshold) & (self._av["scaling"].max(dim=1).values > self.percent_dense * scene_extent)
        stds = self._av["scaling"][sel].repeat(N, 1)
        stds = torch.cat([stds, torch.zeros_like(stds[:, :1])], dim=-1)
        samples = torch.normal(mean=torch.zeros_like(stds), std=stds, generator=generator)
        rots = quat_to_rotmat(self._pv["rotation"][sel]).repeat(N, 1, 1)
        new = self._rows(sel, N)
        new["xyz"] = torch.bmm(rots, samples.unsqueeze(-1)).squeeze(-1) + self._pv["xyz"][sel].repeat(N, 1)
        new["scaling"] = torch.log(self._av["scaling"][sel].repeat(N, 1) / (0.8 * N))
        self.densification_postfix(new)
        prune = torch.cat((sel, torch.zeros(N * int(sel.sum()), device=self.device, dtype=torch.bool)))
        self.prune_points(prune)

    def densify_and_prune(self, max_grad, min_opacity, extent, max_screen_size, generator=None):
        """generator: RNG of the split samples (None = global RNG like the reference; view-parallel training passes an identically
        seeded generator on every rank so that the replicas stay identical)."""
        grads = self.xyz_gradient_accum / self.denom
        grads[grads.isnan()] = 0.0
        self.densify_and_clone(grads, max_grad, extent)
        self.densify_and_split(grads, max_grad, extent, generator=generator)
        prune = (self._av["opacity"] < min_opacity).squeeze()
        if max_screen_size:
            prune = prune | (self.max_radii2D > max_screen_size) | (self._av["scaling"].max(dim=1).values > 0.1 * extent)
        self.prune_points(prune)

    def reset_opacity(self):
        """opacity <- min(opacity, 0.01), Adam moments of the opacity group zeroed (scene/gaussian_model.py:209-212,257-270)."""
        new = inverse_sigmoid(torch.min(self._av["opacity"], torch.ones_like(self._av["opacity"]) * 0.01))
        self._pv["opacity"].copy_(new)
        if self.m is not None:
            _views(self.m, self.P)["opacity"].zero_(); _views(self.v, self.P)["opacity"].zero_()
        self.refresh_activations()

    # ------------------------------------------------------------------ checkpoints (train.py:142-144, gaussian_model.py:58-92)
    def _group_tensors(self, buf):
        v = _views(buf, self.P)
        sh = v["sh"].view(self.P, 16, 3)
        return [v["xyz"], sh[:, :1], sh[:, 1:self.n_coef], v["opacity"], v["scaling"], v["rotation"]]

    def capture(self):
        """The reference's checkpoint tuple; the optimiser entry has torch.optim.Adam's state_dict layout."""
        state = {}
        if self.m is not None and self.step_count > 0:
            for i, (mt, vt) in enumerate(zip(self._group_tensors(self.m), self._group_tensors(self.v))):
                state[i] = dict(step=torch.tensor(float(self.step_count)), exp_avg=mt.clone(), exp_avg_sq=vt.clone())
        groups = [dict(lr=self.lr[i] if self.lr else 0.0, name=n, betas=self.betas, eps=self.eps, weight_decay=0, amsgrad=False, params=[i])
                  for i, n in enumerate(GROUPS)]
        return (self.active_sh_degree, self._xyz.clone(), self._features_dc.clone(), self._features_rest.clone(), self._scaling.clone(),
                self._rotation.clone(), self._opacity.clone(), self.max_radii2D, self.xyz_gradient_accum, self.denom,
                dict(state=state, param_groups=groups), self.spatial_lr_scale)

    def restore(self, model_args, training_args):
        (self.active_sh_degree, xyz, f_dc, f_rest, scaling, rotation, opacity, max_radii2D, accum, denom, opt_dict,
         self.spatial_lr_scale) = model_args
        self.set_parameters(xyz.detach(), f_dc.detach(), f_rest.detach(), opacity.detach(), scaling.detach(), rotation.detach())
        self.training_setup(training_args)
        self.max_radii2D = max_radii2D.to(self.device); self.xyz_gradient_accum = accum.to(self.device); self.denom = denom.to(self.device)
        by_name = {g["name"]: g for g in opt_dict["param_groups"]}
        mt, vt = self._group_tensors(self.m), self._group_tensors(self.v)
        for i, n in enumerate(GROUPS):
            st = opt_dict["state"].get(by_name[n]["params"][0])
            if st is not None:
                mt[i].copy_(st["exp_avg"].reshape(mt[i].shape)); vt[i].copy_(st["exp_avg_sq"].reshape(vt[i].shape))
                self.step_count = int(float(st["step"]))
            self.lr[i] = by_name[n]["lr"]

    # ------------------------------------------------------------------ point_cloud.ply (scene/gaussian_model.py:176-255)
    def construct_list_of_attributes(self):
        names = ["x", "y", "z", "nx", "ny", "nz"] + ["f_dc_%d" % i for i in range(3)] + ["f_rest_%d" % i for i in range(3 * (self.n_coef - 1))]
        return names + ["opacity", "scale_0", "scale_1"] + ["rot_%d" % i for i in range(4)]

    def save_ply(self, path):
        import surfel_io
        P = self.P
        xyz = self._xyz.detach().cpu().numpy()
        f_dc = self._features_dc.detach().transpose(1, 2).flatten(start_dim=1).contiguous().cpu().numpy()
        f_rest = self._features_rest.detach().transpose(1, 2).flatten(start_dim=1).contiguous().cpu().numpy()
        cols = np.concatenate((xyz, np.zeros_like(xyz), f_dc, f_rest, self._opacity.detach().cpu().numpy(),
                               self._scaling.detach().cpu().numpy(), self._rotation.detach().cpu().numpy()), axis=1)
        os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
        surfel_io.write_ply(path, self.construct_list_of_attributes(), cols.astype(np.float32))
        return P

    def load_ply(self, path):
        import surfel_io
        props = surfel_io.read_ply(path)
        col = lambda n: np.asarray(props[n], np.float32)
        xyz = np.stack([col("x"), col("y"), col("z")], axis=1)
        f_dc = np.stack([col("f_dc_%d" % i) for i in range(3)], axis=1)[:, None, :]                     # [P,1,3]
        rest_names = sorted((n for n in props if n.startswith("f_rest_")), key=lambda s: int(s.split("_")[-1]))
        n_rest = self.n_coef - 1
        if len(rest_names) != 3 * n_rest:        # scene/gaussian_model.py:231: assert len(extra_f_names) == 3*(max_sh_degree + 1)**2 - 3
            raise ValueError("%s holds %d f_rest properties, expected %d (sh_degree %d)" % (path, len(rest_names), 3 * n_rest, self.max_sh_degree))
        f_rest = (np.stack([col(n) for n in rest_names], axis=1).reshape(-1, 3, n_rest).transpose(0, 2, 1) if n_rest
                  else np.zeros((xyz.shape[0], 0, 3), np.float32))                                       # channel-major -> [P,n_rest,3]
        scale_names = sorted((n for n in props if n.startswith("scale_")), key=lambda s: int(s.split("_")[-1]))
        rot_names = sorted((n for n in props if n.startswith("rot")), key=lambda s: int(s.split("_")[-1]))
        scales = np.stack([col(n) for n in scale_names], axis=1)
        rots = np.stack([col(n) for n in rot_names], axis=1)
        self.set_parameters(xyz, f_dc, np.ascontiguousarray(f_rest), col("opacity")[:, None], scales, rots)
        self.active_sh_degree = self.max_sh_degree
