"""Training driver on MI355X: the loop of /root/reference/train.py:31-144 on the fused kernels, with the
view-parallel multi-GPU step of surfel_dist.py.

Per iteration (1 view per GPU):  rasterize -> photometric loss (1 fwd + 1 bwd kernel) -> regularisers straight from allmap
(1 fwd + 1 bwd kernel) -> rasterizer backward writing into the flat gradient store -> densification statistics (1 kernel)
-> [RCCL: one all-reduce of the 40 B/surfel geometry gradients + one all-gather of 12 B/surfel/rank colour gradients, from which
the 192 B/surfel SH gradients are rebuilt locally] -> fused Adam + activations (2 kernels).  The reference spends ~150 small PyTorch
kernels on the same work around its rasterizer.

Semantics kept from the reference: learning-rate schedule, SH degree every 1000 iterations, lambda_dist after 3000 and
lambda_normal after 7000 iterations, densify/prune/opacity-reset schedule, no optimiser update on the iterations that
re-create the parameters (densification), Adam(eps=1e-15).  Multi-GPU: G views per step with averaged gradients; the
densification statistics are accumulated locally and all-reduced only when a densification is due.
"""
import math
import os
import random
import time
from types import SimpleNamespace

import numpy as np
import torch
import torch.distributed as dist

import surfel_dist
import surfel_native as _n
from surfel_losses import scalars_from_band_sums, train_loss, train_loss_band, train_loss_manual, train_loss_manual_backward
from surfel_model import COLOUR_FLOATS, GEOM_FLOATS, GaussianModel, exchange_collectives, exchange_same_view
from surfel_render import Camera, post_consts_rows, rasterize, rasterize_manual, rasterize_manual_backward, render


def optimization_params(**over):
    """Defaults of arguments/__init__.py:75-95 (OptimizationParams)."""
    d = dict(iterations=30_000, position_lr_init=0.00016, position_lr_final=0.0000016, position_lr_delay_mult=0.01,
             position_lr_max_steps=30_000, feature_lr=0.0025, opacity_lr=0.05, scaling_lr=0.005, rotation_lr=0.001,
             percent_dense=0.01, lambda_dssim=0.2, lambda_dist=0.0, lambda_normal=0.05, opacity_cull=0.05,
             densification_interval=100, opacity_reset_interval=3000, densify_from_iter=500, densify_until_iter=15_000,
             densify_grad_threshold=0.0002, dist_from_iter=3000, normal_from_iter=7000,
             sh_degree_interval=1000)          # train.py:61-62 hard-codes the 1000
    d.update(over)
    return SimpleNamespace(**d)


def scale_schedule(opt, n, lr="linear"):
    """View-parallel training sees n views per step.  This returns the schedule with every iteration count divided by n (total
    iterations, position-lr horizon, densification window / interval, opacity reset, regulariser and SH-degree onsets), so that
    the run consumes the SAME number of images as the reference's 1-view-per-step schedule and every event happens after the same
    number of images, and with the learning rates scaled: lr = "linear" (x n, default), "sqrt" (x sqrt n) or "none".
    PSNR-parity experiment (profiles/r02_psnr_parity.md, N = 2 / 4 / 8): "linear" keeps the train PSNR of the 1-view reference
    run (+0.05 ... +1.0 dB) in 1/n of the steps; "none" loses 3 - 4 dB at N >= 4; keeping the step count instead (no call to this
    function, gradients averaged) is never below the reference (+1.9 ... +3.4 dB: it sees n x the images)."""
    n = max(1, int(n))
    d = dict(vars(opt))
    f = {"linear": float(n), "sqrt": float(n) ** 0.5, "none": 1.0}[lr]
    for k in ("position_lr_init", "position_lr_final", "feature_lr", "opacity_lr", "scaling_lr", "rotation_lr"):
        d[k] = d[k] * f
    for k in ("iterations", "position_lr_max_steps", "densification_interval", "opacity_reset_interval", "densify_from_iter",
              "densify_until_iter", "dist_from_iter", "normal_from_iter", "sh_degree_interval"):
        d[k] = max(1, int(round(d[k] / n)))
    return SimpleNamespace(**d)


def pipeline_params(**over):
    """Defaults of arguments/__init__.py:66-72 (PipelineParams)."""
    d = dict(convert_SHs_python=False, compute_cov3D_python=False, depth_ratio=0.0, debug=False)
    d.update(over)
    return SimpleNamespace(**d)


def psnr(img1, img2):
    """utils/image_utils.py:19-21."""
    mse = ((img1 - img2) ** 2).reshape(img1.shape[0], -1).mean(1, keepdim=True)
    return 20 * torch.log10(1.0 / torch.sqrt(mse))


# ------------------------------------------------------------------------------------------------ synthetic captures
def look_at(eye, target=(0.0, 0.0, 0.0), up=(0.0, -1.0, 0.0)):
    """(R, T) in the reference's Camera convention: R = C2W rotation (columns = camera x right, y down, z forward), T = W2C translation."""
    eye = np.asarray(eye, np.float64); target = np.asarray(target, np.float64)
    fwd = target - eye; fwd /= np.linalg.norm(fwd)
    right = np.cross(fwd, np.asarray(up, np.float64)); right /= np.linalg.norm(right)
    down = np.cross(fwd, right)
    R = np.stack([right, down, fwd], axis=1)
    return R, -R.T @ eye


def orbit_cameras(n_views, W, H, radius=4.0, fov_deg=50.0, device="cuda", seed=0):
    """Cameras on a sphere of `radius` looking at the origin (NeRF-synthetic style), images filled in later."""
    rng = np.random.default_rng(seed)
    fovx = math.radians(fov_deg)
    fovy = 2 * math.atan(math.tan(fovx / 2) * H / W)
    cams = []
    for i in range(n_views):
        az = 2 * math.pi * (i + 0.5 * rng.uniform()) / n_views
        el = math.radians(rng.uniform(-25, 35))
        eye = radius * np.array([math.cos(el) * math.cos(az), -math.sin(el), math.cos(el) * math.sin(az)])
        R, T = look_at(eye)
        cams.append(Camera(colmap_id=i, R=R, T=T, FoVx=fovx, FoVy=fovy, image=torch.zeros(3, H, W), image_name="syn_%03d" % i, uid=i,
                           data_device=device))
    return cams


def cameras_extent(cams):
    """Scene radius as scene/dataset_readers.py:40-64 (getNerfppNorm) computes it: 1.1 x max distance of a camera from the mean."""
    c = torch.stack([cam.camera_center for cam in cams]).double().cpu().numpy()
    return float(np.linalg.norm(c - c.mean(0, keepdims=True), axis=1).max() * 1.1)


def synthetic_object(P, device, seed=0, extent=1.2, px_scale=0.03):
    """A ground-truth surfel set: a thick random shell of oriented discs around the origin (raw, pre-activation parameters)."""
    g = torch.Generator().manual_seed(seed)
    d = torch.randn((P, 3), generator=g); d = d / d.norm(dim=1, keepdim=True)
    r = extent * (0.6 + 0.4 * torch.rand((P, 1), generator=g))
    xyz = d * r
    # discs roughly tangent to the shell: rotate z-axis onto the radial direction
    z = torch.tensor([0.0, 0.0, 1.0]).expand(P, 3)
    axis = torch.cross(z, d, dim=1); s = axis.norm(dim=1, keepdim=True).clamp_min(1e-8); axis = axis / s
    ang = torch.atan2(s.squeeze(1), (z * d).sum(1))
    quat = torch.cat([torch.cos(ang / 2)[:, None], axis * torch.sin(ang / 2)[:, None]], dim=1)
    scaling = torch.log(px_scale * torch.exp(0.4 * torch.randn((P, 2), generator=g)))
    opacity = torch.logit(torch.rand((P, 1), generator=g) * 0.5 + 0.45)
    f_dc = torch.randn((P, 1, 3), generator=g) * 0.8
    f_rest = torch.randn((P, 15, 3), generator=g) * 0.05
    m = GaussianModel(3, device=device)
    m.set_parameters(xyz, f_dc, f_rest, opacity, scaling, quat)
    m.active_sh_degree = 3
    return m


def capture_views(gt_model, cams, background, pipe=None):
    """Fill the cameras' original_image with renders of the ground-truth surfels (the synthetic stand-in for a dataset)."""
    pipe = pipe or pipeline_params()
    with torch.no_grad():
        for cam in cams:
            img = render(cam, gt_model, pipe, background)["render"]
            cam.original_image = img.clamp(0.0, 1.0).contiguous()
    return cams


# ------------------------------------------------------------------------------------------------ the loop
class Trainer:
    def __init__(self, model, cams, opt=None, pipe=None, white_background=False, extent=None, seed=0, sharding="views", first_iter=0,
                 rehearse_exchange=False, solo=False):
        """sharding (N > 1): "views" = every rank trains on its own view per step (default, BASELINE config 4); "bands" = all ranks
        render row bands of the SAME view (tile-band sharding, BASELINE config 5): every rank evaluates the loss on ITS band plus a
        32-row halo received from its two neighbours (surfel_losses.train_loss_band), back-propagates its own rows, and the
        per-surfel gradients of the bands ADD UP to the single-GPU gradient — ONE all-reduce of 52 B/surfel (geometry + colour
        gradients; the camera is shared, so the SH gradients are rebuilt from the SUM of the colour gradients), no averaging.
        Band edges follow the previous frames' instances per tile row (re-balanced every `rebalance_every` iterations).
        rehearse_exchange: take the view-parallel step (schedule, collectives, split optimiser) even with a single rank — the
        N > 1 code path run against the real backend on one GPU (tests/test_gpu_train.py, scripts/rccl_selfcheck.py).
        solo: ignore an initialised process group — this rank trains alone (bench.py: every rank prepares the same trained state by
        itself, deterministically, before the timed view-parallel Trainer synchronises the replicas)."""
        if sharding not in ("views", "bands"):
            raise ValueError("sharding must be 'views' or 'bands'")
        self.sharding = sharding
        self.model, self.cams = model, cams
        self.opt = opt or optimization_params()
        self.pipe = pipe or pipeline_params()
        self.white_background = white_background
        self.background = torch.tensor([1.0, 1.0, 1.0] if white_background else [0.0, 0.0, 0.0], dtype=torch.float32, device=model.device)
        self.extent = extent if extent is not None else cameras_extent(cams)
        self.world = dist.get_world_size() if (not solo and dist.is_available() and dist.is_initialized()) else 1
        self.rank = dist.get_rank() if self.world > 1 else 0
        self.seed = seed
        self._rng = random.Random(seed)
        self._stack = []
        self.iteration = int(first_iter)      # resume: the checkpoint's iteration (train.py:37-39), so that the lr / SH / densify schedules continue
        self.last = {}
        self._epoch, self._epoch_views, self._epoch_campos, self._centers = -1, None, None, None
        self._one = torch.ones((), dtype=torch.float32, device=model.device)
        self.defer_scalars = os.environ.get("SURFEL_DEFER_SCALARS", "1") != "0"      # loss scalars by the fused loss backward launch instead of a finalize launch
        self.lazy_count = os.environ.get("SURFEL_LAZY_COUNT", "1") != "0"      # forward without the host wait for the instance count (step())
        self.lazy_overflows = 0
        self.manual_chain = os.environ.get("SURFEL_MANUAL_CHAIN", "1") != "0"      # forward / backward of the iteration driven without autograd (step())
        self.views_per_step = 1         # single process: > 1 = accumulate that many views per optimiser step (_step_accumulate)
        self.rebalance_every = 8        # bands: iterations between re-balancing the band edges (one small all-gather + D2H)
        self._row_weights = None        # bands: running mean of tile instances per 16-row tile row (host list) of frames of ...
        self._row_weights_hw = None     # ... this (height, width): cameras of another resolution start a fresh mean
        self.wire = {"total": 0}        # bytes on the wire per GPU of the most recent step, by collective (surfel_dist.wire_bytes_per_step)
        self.time_exchange = False      # bench: bracket the stream waits on the exchange with events -> self.exchange_events
        self.exchange_events = []
        # RCCL: asynchronous collectives with stream-level waits (SH-block Adam overlaps the geometry all-reduce); other backends
        # (gloo rehearsals stage through the host and block) take the plain synchronous form
        self._exchange = self.world > 1 or (bool(rehearse_exchange) and dist.is_available() and dist.is_initialized())
        self._async_exchange = self._exchange and dist.get_backend() == "nccl"
        # views + RCCL: the all-gather of the colour gradients starts inside the rasterizer's backward, as soon as the kernel that
        # finalises them is on the stream (surfel_set_backward_hook), and overlaps the per-surfel geometry chain rule.  The split
        # costs a second pass over the gradient records (measured +15 us at C2, +208 us at C4: scripts/hook_cost.py) and buys up to
        # min(gather time, chain-rule time): worth it once the gather is long, i.e. from 4 ranks on (12 B/surfel/rank received)
        # ... or not: RCCL's launch latency, the link count and the frame size decide, so with more than one rank it is MEASURED at
        # start-up ("auto": four iterations with the early gather, four without, device time by events on the compute stream, the
        # slower rank decides for all: _probe_early_gather) — results are bit-identical either way.
        self.early_gather = True if rehearse_exchange else ("auto" if (self.world > 1 and self._async_exchange and sharding == "views") else False)
        self.early_gather_probe = None      # {"early_ms": .., "late_ms": .., "choice": ..} once decided
        self._eg_events, self._eg_first, self._eg_failed = [], None, False
        self._early, self._early_err = None, None
        # fused SH path (default): the rasterizer's backward skips the 192 B/surfel SH gradients, the optimiser kernel rebuilds them
        # from the 12 B/surfel colour gradients.  Always on under view-parallel training (that is how the gradients are exchanged).
        self.fused_sh = self._exchange or os.environ.get("SURFEL_SH_FUSED", "1") != "0"
        self.fused_update = os.environ.get("SURFEL_FUSED_UPDATE", "1") != "0"      # surfel_train_update (statistics + Adam in one launch)
        if model.grad is None:
            model.training_setup(self.opt)
        if self.world > 1:
            self._sync_replicas()

    def _sync_replicas(self):
        """N > 1: every rank must start from the same model.  Surfel count and optimiser step are checked (a mismatch would make
        the fixed-size collectives hang or corrupt memory), then parameters, Adam moments and densification statistics are
        broadcast from rank 0 (create_from_pcd draws its rotations from the global RNG, so differently seeded ranks differ)."""
        m = self.model
        chk = torch.tensor([m.P, m.step_count, self.iteration], dtype=torch.int64, device=m.device)
        lo, hi = chk.clone(), chk.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN); dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        if not torch.equal(lo, hi):
            raise RuntimeError("view-parallel ranks disagree on (surfels, optimiser step, iteration): min %s max %s" % (lo.tolist(), hi.tolist()))
        for t in (m.theta, m.m, m.v, m.xyz_gradient_accum, m.denom, m.max_radii2D):
            if t is not None and t.numel():
                dist.broadcast(t, src=0)
        m.refresh_activations()

    def _step_views(self):
        """(camera indices of ALL ranks for the current iteration, their camera centres [world,3] on the device).  Every rank
        computes the same schedule (surfel_dist.view_indices); it is built once per epoch, so a step costs no host RNG work
        and no device gather."""
        per_epoch = max(1, len(self.cams) // self.world)
        epoch, k = divmod(self.iteration - 1, per_epoch)
        if self._epoch != epoch:
            self._epoch = epoch
            self._epoch_views = surfel_dist.epoch_schedule(len(self.cams), self.world, epoch, self.seed)
            if self._centers is None:
                self._centers = torch.stack([c.camera_center for c in self.cams]).contiguous()
            idx = torch.tensor(self._epoch_views, dtype=torch.long, device=self._centers.device)
            self._epoch_campos = self._centers[idx.reshape(-1)].reshape(per_epoch, self.world, 3).contiguous()
        return self._epoch_views[k], self._epoch_campos[k]

    def _next_camera(self):
        if self._exchange and self.sharding == "views":
            return self.cams[self._step_views()[0][self.rank]]
        if not self._stack:                       # train.py:64-67: pop a random view from a refilled stack
            self._stack = list(range(len(self.cams)))
        return self.cams[self._stack.pop(self._rng.randint(0, len(self._stack) - 1))]

    def _reduce_stats(self):
        m = self.model
        packed = torch.cat([m.xyz_gradient_accum, m.denom], dim=1).contiguous()
        dist.all_reduce(packed, op=dist.ReduceOp.SUM)
        m.xyz_gradient_accum, m.denom = packed[:, :1].contiguous(), packed[:, 1:2].contiguous()
        dist.all_reduce(m.max_radii2D, op=dist.ReduceOp.MAX)

    def step(self):
        """One training iteration (train.py:54-138).  Returns nothing; `self.last` holds device scalars for logging."""
        if self.world == 1 and self.views_per_step > 1:
            return self._step_accumulate(self.views_per_step)
        self.iteration += 1
        it, opt, m = self.iteration, self.opt, self.model
        m.update_learning_rate(it)
        if it % getattr(opt, "sh_degree_interval", 1000) == 0:
            m.oneupSHdegree()
        cam = self._next_camera()
        m.bind(sh_grad=not self.fused_sh)      # fused: the SH gradients are rebuilt inside the optimiser kernel from the colour gradients
        bands = self.world > 1 and self.sharding == "bands"
        lam_n = opt.lambda_normal if it > opt.normal_from_iter else 0.0
        lam_d = opt.lambda_dist if it > opt.dist_from_iter else 0.0
        reg = lam_n > 0.0 or lam_d > 0.0
        stats_live = it < opt.densify_until_iter
        if bands:
            H, W = int(cam.image_height), int(cam.image_width)
            if self._row_weights_hw != (H, W):      # real capture sets mix resolutions: the running weights belong to one of them
                self._row_weights, self._row_weights_hw = None, (H, W)
            bounds = surfel_dist.band_bounds(H, self.world, self._row_weights, multiple=surfel_dist.HALO)
            y0, y1 = bounds[self.rank]
            # the densification statistic of the view = norm of the SUM over bands: it rides in the same all-reduce, right
            # behind the colour block (both live in the SH section of the gradient store, unused in fused-SH mode)
            arena2d = m.grad[(GEOM_FLOATS + COLOUR_FLOATS) * m.P:(GEOM_FLOATS + COLOUR_FLOATS + 3) * m.P].view(m.P, 3)
            import diff_surfel_rasterization as dsr
            dsr._grad_arena["means2D"] = arena2d
            image_b, radii, allmap_b, means2D = rasterize(cam, m, self.pipe, self.background, zero_means2D=False, band=(y0, y1))
            ext = surfel_dist.exchange_halo(torch.cat([image_b, allmap_b], 0) if reg else image_b, bounds, H)
            top, bot = surfel_dist.halo_rows(bounds, self.rank, H)
            gt_ext = cam.original_image[:, y0 - top:y1 + bot]
            consts = post_consts_rows(cam.post_consts(), y0 - top) if reg else None
            loss, sums = train_loss_band(ext[:3], ext[3:] if reg else None, gt_ext, consts, self.pipe.depth_ratio, opt.lambda_dssim, lam_n, lam_d,
                                         (top, top + (y1 - y0)), (H, W))
            sums = sums.clone()
            dist.all_reduce(sums, op=dist.ReduceOp.SUM)           # 16 bytes: the full-image loss terms, for logging
            scalars = scalars_from_band_sums(sums, float(3 * H * W), float(H * W), opt.lambda_dssim, lam_n, lam_d)
            halo_b = surfel_dist.halo_bytes(bounds, self.rank, H, W, 10 if reg else 3)
        else:
            halo_b = 0
        self.wire = surfel_dist.wire_bytes_per_step(m.P, self.world, self.sharding, stats_live, halo_b)
        early = (not bands) and self._probe_early_gather() and self._async_exchange and it < opt.iterations
        # Lazily counted forward (include/surfel_hip.h: SURFEL_OPT_LAZY_COUNT): the host does not wait for the frame's instance count in
        # the middle of the iteration — it enqueues loss and backward behind the forward and collects the count afterwards, when it has
        # long arrived; a frame that overflowed its binning capacity (rare) is rendered again with exact sizes, loss and backward with
        # it, before anything irreversible (statistics, optimiser step, collectives) has happened.
        lazy = self.lazy_count and not bands and not early
        # The iteration's chain is fixed (rasterizer -> loss -> loss backward -> rasterizer backward): driven by hand
        # (surfel_native.ManualCtx) it costs a fraction of the host time autograd spends on it — engine, worker-thread hand-over, five
        # parameter gates — with the same kernels and bits; compute_cov3D_python trains through PyTorch code and needs autograd.
        manual = self.manual_chain and not bands and not getattr(self.pipe, "compute_cov3D_python", False)
        g2d = None
        for bits in ((_n.OPT_LAZY_COUNT, _n.OPT_EXACT_BINNING) if lazy else (0,)):
            if manual:
                with torch.no_grad():
                    rctx, image, radii, allmap = rasterize_manual(cam, m, self.pipe, self.background, debug_bits=bits)
                    lctx, loss, scalars = train_loss_manual(image, allmap if reg else None, cam.original_image, cam.post_consts() if reg else None,
                                                            self.pipe.depth_ratio, opt.lambda_dssim, lam_n, lam_d, defer_scalars=self.defer_scalars)
            elif not bands:
                image, radii, allmap, means2D = rasterize(cam, m, self.pipe, self.background, zero_means2D=False, debug_bits=bits)
                loss, scalars = train_loss(image, allmap if reg else None, cam.original_image, cam.post_consts() if reg else None,
                                           self.pipe.depth_ratio, opt.lambda_dssim, lam_n, lam_d, defer_scalars=self.defer_scalars)      # (read after the backward below)
            if early:
                self._early, self._early_err = None, None
                _n.set_backward_hook(self._on_colour_ready)
            try:
                if manual:
                    with torch.no_grad():
                        g_img, g_am = train_loss_manual_backward(lctx, self._one)
                        g2d = rasterize_manual_backward(rctx, g_img, g_am)
                else:
                    torch.autograd.backward(loss, grad_tensors=self._one)        # cached seed gradient: no ones_like fill per iteration
            finally:
                if early:
                    _n.set_backward_hook(None)
            if not lazy:
                break
            try:
                import diff_surfel_rasterization as dsr
                dsr.finish_count()
                break
            except _n.CapacityOverflow:
                self.lazy_overflows += 1
                lazy = False      # second pass: exact binning, count known when the forward returns
        if self._early_err is not None:
            # the early all-gather could not be launched from inside the backward: dL/dcolour is final regardless (the split
            # kernel ran), so the step continues with the gather-after-backward form and the early form stays off
            import warnings
            warnings.warn("early colour all-gather disabled after: %r" % (self._early_err,))
            if self.early_gather == "auto":
                # mid-probe: the other ranks still expect this rank in the verdict's all-reduce (_probe_early_gather) — keep the state
                # machine running, take the late form from here on, and vote "early = never" when the verdict is due
                self._eg_failed = True
            else:
                self.early_gather = False
            self._early, self._early_err = None, None
        self.last = dict(loss=scalars[5], scalars=scalars, points=m.P, radii=radii)     # [Ll1, ssim, normal_err, dist, photometric, total] on the device
        with torch.no_grad():
            rebuilt = False
            w_same = None
            if bands:
                # ONE all-reduce: geometry 10 | colour 3 (| means2D statistic 3) floats per surfel
                n_f = GEOM_FLOATS + COLOUR_FLOATS + (3 if stats_live else 0)
                w_same = dist.all_reduce(m.grad[:n_f * m.P], op=dist.ReduceOp.SUM, async_op=self._async_exchange)
                if stats_live:
                    radii = radii.clone(); dist.all_reduce(radii, op=dist.ReduceOp.MAX)      # visibility = union of the bands
                if self._async_exchange:
                    self._timed_wait(w_same)
                self._rebalance_bands(cam)
            # statistics + optimiser step as ONE launch where nothing can come between them (single GPU, SH block rebuilt in the kernel,
            # no densification / opacity reset this iteration): two launch boundaries and the statistics' latency-bound kernel fewer
            one_launch = self.fused_update and self.fused_sh and not bands and not self._exchange and it < opt.iterations and not self._is_event_iteration(it)
            if one_launch:
                m.update_step((cam.camera_center[None], m.gcol[None]), stats=((g2d if manual else means2D.grad), radii) if stats_live else None)
                rebuilt = True      # (nothing left to do below)
            elif stats_live:
                m.add_densification_stats(arena2d if bands else (g2d if manual else means2D.grad), radii=radii)
                rebuilt = self._schedule_events(it, bands)
            if it < opt.iterations and not rebuilt:     # re-created parameters carry no gradient in the reference: no update
                if bands:
                    # partial gradients of one view add up (no averaging); the SH block is rebuilt from the summed colour gradients
                    m.optimizer_step(grad_scale=1.0, colour_grads=(cam.camera_center[None], m.gcol[None]))
                elif self._exchange:
                    # all-reduce of the 40 B/surfel geometry prefix + all-gather of 12 B/surfel/rank colour gradients; the 192 B/surfel
                    # SH gradients are rebuilt from them (exact, rank-ordered sum) instead of being all-reduced
                    campos_all = self._step_views()[1]
                    scale = 1.0 / self.world      # views: average
                    if self._async_exchange:
                        if self._early is not None:      # the gather has been in flight since the middle of the backward
                            gcol_all, w_gather = self._early
                            self._early = None
                            w_reduce = dist.all_reduce(m.grad[:GEOM_FLOATS * m.P], op=dist.ReduceOp.SUM, async_op=True)
                        else:
                            gcol_all, w_gather, w_reduce = exchange_collectives(m.grad, m.gcol, m.P, async_op=True)
                        self._timed_wait(w_gather)   # stream-level wait: the SH block updates while the geometry all-reduce is in flight
                        m.optimizer_step(grad_scale=scale, colour_grads=(campos_all, gcol_all), parts=1)
                        self._timed_wait(w_reduce)
                        m.optimizer_step(grad_scale=scale, colour_grads=(campos_all, gcol_all), parts=2)
                    else:
                        gcol_all = exchange_collectives(m.grad, m.gcol, m.P)
                        m.optimizer_step(grad_scale=scale, colour_grads=(campos_all, gcol_all))
                else:
                    if self.fused_sh:
                        campos_all, gcol_all = cam.camera_center[None], m.gcol[None]
                    m.optimizer_step(grad_scale=1.0, colour_grads=(campos_all, gcol_all) if self.fused_sh else None)
            if self._early is not None:      # an iteration without an optimiser step (parameters re-created): retire the gather
                self._early[1].wait()
                self._early = None

    EG_WARM, EG_LEN = 2, 4      # start-up probe: iterations skipped, iterations per setting

    @staticmethod
    def decide_early_gather(early_ms, late_ms):
        """The early gather pays a second pass over the gradient records: keep it only when it wins by more than measurement noise."""
        return early_ms < 0.98 * late_ms

    def _probe_early_gather(self):
        """early_gather == "auto": whether THIS iteration takes the early gather; after 2 x EG_LEN probe iterations the faster setting
        (max over ranks of the device time) is fixed for the rest of the run."""
        if self.early_gather != "auto":
            return bool(self.early_gather)
        if self._eg_first is None:
            # the two windows must time the same kind of iteration: start behind the warm-up at the first iteration from which
            # 2 x EG_LEN iterations hold no densification / opacity reset (a function of the iteration number: the same on every rank)
            first = self.iteration + self.EG_WARM
            while any(self._is_event_iteration(first + k) for k in range(2 * self.EG_LEN + 1)):
                first += 1
            self._eg_first = first
        ph = self.iteration - self._eg_first
        if ph < 0:
            return False
        if ph in (0, self.EG_LEN, 2 * self.EG_LEN):
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            self._eg_events.append(ev)
        if ph < self.EG_LEN:
            return not self._eg_failed      # (a rank whose early gather failed takes the late form: the same collectives, later)
        if ph < 2 * self.EG_LEN:
            return False
        try:
            e0, e1, e2 = self._eg_events
            e2.synchronize()
            t = torch.tensor([e0.elapsed_time(e1), e1.elapsed_time(e2)], dtype=torch.float32, device=self.model.device)
        except Exception:      # noqa: BLE001 — no timing, no early gather; every rank still joins the collective below
            t = torch.tensor([1.0, 0.0], dtype=torch.float32, device=self.model.device)
        if self._eg_failed:      # this rank cannot launch the early gather: MAX over the ranks turns every rank's verdict to "late"
            t = torch.tensor([3.0e38, 0.0], dtype=torch.float32, device=self.model.device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        early_ms, late_ms = (float(x) / self.EG_LEN for x in t.tolist())
        self.early_gather = self.decide_early_gather(early_ms, late_ms)
        self.early_gather_probe = {"early_ms_per_step": round(early_ms, 4), "late_ms_per_step": round(late_ms, 4), "choice": "early" if self.early_gather else "late"}
        self._eg_events = []
        return self.early_gather

    def _is_event_iteration(self, it):
        """Does iteration `it` densify / prune or reset the opacities (_schedule_events)?  Those iterations cost more and skip the optimiser step."""
        opt = self.opt
        if it < opt.densify_until_iter and it > opt.densify_from_iter and it % opt.densification_interval == 0:
            return True
        return it < opt.densify_until_iter and (it % opt.opacity_reset_interval == 0 or (self.white_background and it == opt.densify_from_iter))

    def _on_colour_ready(self):
        """Called by the C library inside the rasterizer's backward (autograd's thread, the forward's stream) once dL/dcolour is
        final on the stream: launch the all-gather of this rank's colour gradients now.  Exceptions cannot cross the C frame:
        they are parked and re-raised by step()."""
        try:
            m = self.model
            gall = torch.empty((dist.get_world_size(), m.P, 3), dtype=torch.float32, device=m.device)
            self._early = (gall, dist.all_gather_into_tensor(gall, m.gcol, async_op=True))
        except Exception as e:      # noqa: BLE001
            self._early, self._early_err = None, e

    def _schedule_events(self, it, bands=False):
        """Densification and opacity reset when due (train.py:129-135); returns whether the parameters were re-created."""
        opt, m = self.opt, self.model
        rebuilt = False
        if it > opt.densify_from_iter and it % opt.densification_interval == 0:
            if self.world > 1 and not bands:
                self._reduce_stats()
            size_threshold = 20 if it > opt.opacity_reset_interval else None
            gen = None
            if self.world > 1:       # identical split samples on every rank
                gen = torch.Generator(device=m.device); gen.manual_seed(self.seed * 1_000_003 + it)
            m.densify_and_prune(opt.densify_grad_threshold, opt.opacity_cull, self.extent, size_threshold, generator=gen)
            rebuilt = True
        if it % opt.opacity_reset_interval == 0 or (self.white_background and it == opt.densify_from_iter):
            m.reset_opacity()
            if not rebuilt:      # the reference re-creates only the opacity parameter: its update is skipped this iteration
                m._gv["opacity"].zero_()
        return rebuilt

    def _step_accumulate(self, n):
        """One optimiser step on the AVERAGED gradients of n views rendered one after the other on this GPU — the optimisation
        semantics of n view-parallel ranks (same view schedule: surfel_dist.epoch_schedule; per-view densification statistics;
        SH gradients rebuilt from the n views' colour gradients) without the processes.  Used by the PSNR-parity experiment."""
        self.iteration += 1
        it, opt, m = self.iteration, self.opt, self.model
        m.update_learning_rate(it)
        if it % getattr(opt, "sh_degree_interval", 1000) == 0:
            m.oneupSHdegree()
        per_epoch = max(1, len(self.cams) // n)
        epoch, k = divmod(it - 1, per_epoch)
        if self._epoch != epoch:
            self._epoch, self._epoch_views = epoch, surfel_dist.epoch_schedule(len(self.cams), n, epoch, self.seed)
        views = [self.cams[v] for v in self._epoch_views[k]]
        lam_n = opt.lambda_normal if it > opt.normal_from_iter else 0.0
        lam_d = opt.lambda_dist if it > opt.dist_from_iter else 0.0
        reg = lam_n > 0.0 or lam_d > 0.0
        geo = torch.zeros(GEOM_FLOATS * m.P, device=m.device)
        gcols = torch.empty((n, m.P, 3), device=m.device)
        for r, cam in enumerate(views):
            m.bind(sh_grad=False)
            image, radii, allmap, means2D = rasterize(cam, m, self.pipe, self.background, zero_means2D=False)
            loss, scalars = train_loss(image, allmap if reg else None, cam.original_image, cam.post_consts() if reg else None,
                                       self.pipe.depth_ratio, opt.lambda_dssim, lam_n, lam_d)
            torch.autograd.backward(loss, grad_tensors=self._one)
            with torch.no_grad():
                geo += m.grad[:GEOM_FLOATS * m.P]; gcols[r].copy_(m.gcol)
                if it < opt.densify_until_iter:
                    m.add_densification_stats(means2D.grad, radii=radii)
        self.last = dict(loss=scalars[5], scalars=scalars, points=m.P, radii=radii)
        with torch.no_grad():
            rebuilt = self._schedule_events(it) if it < opt.densify_until_iter else False
            if it < opt.iterations and not rebuilt:
                m.grad[:GEOM_FLOATS * m.P].copy_(geo)
                campos = torch.stack([c.camera_center for c in views]).contiguous()
                m.optimizer_step(grad_scale=1.0 / n, colour_grads=(campos, gcols))

    def _timed_wait(self, work):
        """Stream-level wait on an asynchronous collective; with time_exchange the wait is bracketed by events on the compute
        stream — the elapsed time is what the exchange was NOT hidden behind compute (resolved by the caller after the run)."""
        if self.time_exchange:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); work.wait(); e1.record()
            self.exchange_events.append((e0, e1))
        else:
            work.wait()

    def _rebalance_bands(self, cam):
        """bands: every `rebalance_every` iterations gather the ranks' tile instances per tile row (a few hundred bytes) into a
        running mean that the next iterations' band_bounds use as weights."""
        if self.rebalance_every <= 0 or self.iteration % self.rebalance_every != 0:
            return
        import diff_surfel_rasterization as dsr
        rows16 = (int(cam.image_height) + 15) // 16
        mine = dsr.tile_row_instances()
        full = torch.zeros((rows16,), dtype=torch.int64, device=self.model.device)
        y0 = surfel_dist.band_bounds(int(cam.image_height), self.world, self._row_weights, multiple=surfel_dist.HALO)[self.rank][0]
        full[y0 // 16:y0 // 16 + mine.numel()] = mine
        dist.all_reduce(full, op=dist.ReduceOp.SUM)
        w = full.to(torch.float64).cpu().tolist()
        self._row_weights = w if self._row_weights is None else [0.5 * a + 0.5 * b for a, b in zip(self._row_weights, w)]

    def evaluate(self, cams=None):
        """Mean PSNR / L1 over views (training_report, train.py:201-232)."""
        cams = cams or self.cams
        with torch.no_grad():
            ps, l1 = [], []
            for cam in cams:
                img = torch.clamp(render(cam, self.model, self.pipe, self.background)["render"], 0.0, 1.0)
                gt = torch.clamp(cam.original_image, 0.0, 1.0)
                ps.append(psnr(img, gt).mean()); l1.append((img - gt).abs().mean())
            return float(torch.stack(ps).mean()), float(torch.stack(l1).mean())


def training(model, cams, opt=None, pipe=None, iterations=None, white_background=False, log_every=0, seed=0):
    """Run the loop; returns the Trainer (model trained in place).  log_every > 0 prints loss / points / it/s."""
    opt = opt or optimization_params()
    if iterations is not None:
        opt.iterations = iterations
    tr = Trainer(model, cams, opt, pipe, white_background, seed=seed)
    t0 = time.perf_counter()
    for _ in range(opt.iterations):
        tr.step()
        if log_every and tr.iteration % log_every == 0 and tr.rank == 0:
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            print("[it %d] loss %.5f points %d  %.1f it/s" % (tr.iteration, float(tr.last["loss"]), model.P, tr.iteration / dt), flush=True)
    return tr
