"""Photometric losses of the reference's training loop on MI355X — drop-in for `utils.loss_utils.l1_loss` / `ssim`
(/root/reference/utils/loss_utils.py:23-24, 43-73; used at train.py:72-74) plus the fused form the trainer uses.

Same names, argument meaning and return shapes; the work is done by two LDS-tiled HIP kernels of libsurfel_hip.so
(include/surfel_train.h) instead of five grouped conv2d calls + ~40 elementwise kernels per iteration.
No CPU / PyTorch fallback: CPU tensors raise.
"""
import os

import torch

import surfel_native as _n

_n.load()


def _check(rc, what):
    if rc < 0:
        raise RuntimeError("%s failed (%d): %s" % (what, rc, _n.last_error()))
    return rc


def _planes(img, gt):
    if img.shape != gt.shape or img.dim() < 2:
        raise ValueError("image / target shapes differ: %s vs %s" % (tuple(img.shape), tuple(gt.shape)))
    if img.device.type != "cuda":
        raise RuntimeError("surfel_losses: tensors must live on a HIP device (got %s)" % img.device)
    H, W = int(img.shape[-2]), int(img.shape[-1])
    return img.numel() // (H * W), H, W


class _L1SSIM(torch.autograd.Function):
    """Returns the two means (mean |img-gt|, mean SSIM map) as a [2] tensor; backward is one kernel."""

    @staticmethod
    def forward(ctx, img, gt, window=11):
        planes, H, W = _planes(img, gt)
        x = img.detach().contiguous().float(); y = gt.detach().contiguous().float()
        dev = x.device
        need = img.requires_grad
        lib = _n.load()
        nblk = ((W + 31) // 32) * ((H + 31) // 32)
        dmaps = torch.empty((3, planes, H, W), dtype=torch.float32, device=dev) if need else None
        partials = torch.empty((planes * nblk, 2), dtype=torch.float32, device=dev)
        out = torch.empty((2,), dtype=torch.float32, device=dev)
        s = _n.current_stream_ptr(dev)
        with torch.cuda.device(dev):
            _check(lib.surfel_l1_ssim_forward_w(int(window), planes, H, W, _n.ptr(x), _n.ptr(y), _n.ptr(dmaps), _n.ptr(partials), s),
                   "surfel_l1_ssim_forward")
            _check(lib.surfel_reduce_partials(_n.ptr(partials), 1, planes * nblk, 2, 1.0 / (planes * H * W), _n.ptr(out), s),
                   "surfel_reduce_partials")
        ctx.dims = (planes, H, W)
        ctx.window = int(window)
        ctx.in_shape = tuple(img.shape)
        if need:
            ctx.save_for_backward(x, y, dmaps)
        return out

    @staticmethod
    def backward(ctx, g):
        planes, H, W = ctx.dims
        x, y, dmaps = ctx.saved_tensors
        dev = x.device
        lib = _n.load()
        N = float(planes * H * W)
        g = g.contiguous().float()          # (dL/d mean|.|, dL/d mean S), stays on the device
        grad = torch.empty_like(x)
        with torch.cuda.device(dev):
            _check(lib.surfel_l1_ssim_backward_w(ctx.window, planes, H, W, _n.ptr(x), _n.ptr(y), _n.ptr(dmaps), 1.0 / N, 1.0 / N, _n.ptr(g[0:1]),
                                                 _n.ptr(g[1:2]), _n.ptr(grad), _n.current_stream_ptr(dev)), "surfel_l1_ssim_backward")
        return grad.view(ctx.in_shape), None, None


def _means(img, gt, window=11):
    out = _L1SSIM.apply(img, gt, window)
    return out


def l1_loss(network_output, gt):
    """mean |network_output - gt| (utils/loss_utils.py:23-24)."""
    return _means(network_output, gt)[0]


class _SSIMPerImage(torch.autograd.Function):
    """size_average=False (utils/loss_utils.py:70-73): one mean SSIM per batch element of a [B,C,H,W] input."""

    @staticmethod
    def forward(ctx, img, gt, window=11):
        B, C, H, W = (int(v) for v in img.shape)
        x = img.detach().contiguous().float(); y = gt.detach().contiguous().float()
        dev = x.device
        lib = _n.load()
        nblk = ((W + 31) // 32) * ((H + 31) // 32)
        dmaps = torch.empty((3, B * C, H, W), dtype=torch.float32, device=dev)
        partials = torch.empty((B * C * nblk, 2), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            _check(lib.surfel_l1_ssim_forward_w(int(window), B * C, H, W, _n.ptr(x), _n.ptr(y), _n.ptr(dmaps), _n.ptr(partials),
                                                _n.current_stream_ptr(dev)), "surfel_l1_ssim_forward")
        ctx.dims = (B, C, H, W)
        ctx.window = int(window)
        ctx.save_for_backward(x, y, dmaps)
        return partials.view(B, C * nblk, 2)[:, :, 1].sum(1) / float(C * H * W)

    @staticmethod
    def backward(ctx, g):
        B, C, H, W = ctx.dims
        x, y, dmaps = ctx.saved_tensors
        dev = x.device
        lib = _n.load()
        g = g.contiguous().float()
        grad = torch.empty_like(x)
        s = _n.current_stream_ptr(dev)
        with torch.cuda.device(dev):
            for b in range(B):        # one launch per batch element: its own upstream scalar
                dm = dmaps[:, b * C:(b + 1) * C].contiguous()
                _check(lib.surfel_l1_ssim_backward_w(ctx.window, C, H, W, _n.ptr(x[b]), _n.ptr(y[b]), _n.ptr(dm), 0.0, 1.0 / float(C * H * W), None,
                                                     _n.ptr(g[b:b + 1]), _n.ptr(grad[b]), s), "surfel_l1_ssim_backward")
        return grad, None, None


def ssim(img1, img2, window_size=11, size_average=True):
    """SSIM with the reference's Gaussian window (sigma 1.5) and zero padding (utils/loss_utils.py:43-73): the mean over everything
    (size_average=True, what train.py uses) or one mean per batch element of a [B,C,H,W] input (size_average=False).
    window_size: odd, 3..15 (the reference's default and only call-site value is 11)."""
    window_size = int(window_size)
    if window_size < 3 or window_size > 15 or window_size % 2 == 0:
        raise NotImplementedError("window_size must be odd and in 3..15 (the HIP kernels are instantiated for these radii)")
    if not size_average:
        if img1.dim() != 4:
            raise ValueError("size_average=False needs a [B,C,H,W] input (the reference reduces dims 1..3)")
        if img1.shape != img2.shape or img1.device.type != "cuda":
            raise RuntimeError("surfel_losses.ssim: shapes differ or tensors are not on a HIP device")
        return _SSIMPerImage.apply(img1, img2, window_size)
    return _means(img1, img2, window_size)[1]


class _PhotometricLoss(torch.autograd.Function):
    """(1 - l) * L1 + l * (1 - SSIM) in ONE forward and ONE backward kernel (train.py:72-74)."""

    @staticmethod
    def forward(ctx, img, gt, lambda_dssim):
        planes, H, W = _planes(img, gt)
        x = img.detach().contiguous().float(); y = gt.detach().contiguous().float()
        dev = x.device
        lib = _n.load()
        nblk = ((W + 31) // 32) * ((H + 31) // 32)
        dmaps = torch.empty((3, planes, H, W), dtype=torch.float32, device=dev)
        partials = torch.empty((planes * nblk, 2), dtype=torch.float32, device=dev)
        means = torch.empty((2,), dtype=torch.float32, device=dev)
        s = _n.current_stream_ptr(dev)
        with torch.cuda.device(dev):
            _check(lib.surfel_l1_ssim_forward(planes, H, W, _n.ptr(x), _n.ptr(y), _n.ptr(dmaps), _n.ptr(partials), s), "surfel_l1_ssim_forward")
            _check(lib.surfel_reduce_partials(_n.ptr(partials), 1, planes * nblk, 2, 1.0 / (planes * H * W), _n.ptr(means), s),
                   "surfel_reduce_partials")
        ctx.dims = (planes, H, W, float(lambda_dssim))
        ctx.in_shape = tuple(img.shape)
        ctx.save_for_backward(x, y, dmaps)
        ctx.mark_non_differentiable(means)
        loss = (1.0 - lambda_dssim) * means[0] + lambda_dssim * (1.0 - means[1])
        return loss, means

    @staticmethod
    def backward(ctx, g_loss, g_means):
        planes, H, W, lam = ctx.dims
        x, y, dmaps = ctx.saved_tensors
        dev = x.device
        N = float(planes * H * W)
        grad = torch.empty_like(x)
        g = g_loss.contiguous().float().reshape(1)
        with torch.cuda.device(dev):
            _check(_n.load().surfel_l1_ssim_backward(planes, H, W, _n.ptr(x), _n.ptr(y), _n.ptr(dmaps), (1.0 - lam) / N, -lam / N, _n.ptr(g),
                                                     _n.ptr(g), _n.ptr(grad), _n.current_stream_ptr(dev)), "surfel_l1_ssim_backward")
        return grad.view(ctx.in_shape), None, None


def photometric_loss(image, gt_image, lambda_dssim=0.2):
    """loss = (1 - lambda_dssim) * l1_loss + lambda_dssim * (1 - ssim)   (train.py:72-74), fused.
    Returns (loss, means) with means = [Ll1, ssim] (detached, device) for logging."""
    return _PhotometricLoss.apply(image, gt_image, lambda_dssim)


# the two halves of the training loss (L1 + SSIM | allmap regularisers) in one launch per direction (csrc/train_fused.hip); 0: separate
# launches — same kernels' bodies, same bits (tests/test_gpu_train.py::test_fused_loss_launches_keep_the_bits)
FUSED_LOSS = os.environ.get("SURFEL_FUSED_LOSS", "1") != "0"


class _TrainLoss(torch.autograd.Function):
    """The whole loss of a training iteration (train.py:72-88) as one autograd node:
    (1-l)*L1 + l*(1-SSIM) + lambda_normal*mean(1 - rend_normal.surf_normal) + lambda_dist*mean(rend_dist)
    forward = SSIM kernel [+ post-processing kernel] + one finalize launch; backward = one kernel per input.
    Returns (total, scalars) with scalars = [Ll1, ssim, normal_err, dist, photometric, total] (detached)."""

    @staticmethod
    def forward(ctx, image, allmap, gt, cam, depth_ratio, lambda_dssim, lambda_normal, lambda_dist, defer_scalars=False):
        planes, H, W = _planes(image, gt)
        x = image.detach().contiguous().float(); y = gt.detach().contiguous().float()
        dev = x.device
        lib = _n.load()
        reg = (lambda_normal != 0.0 or lambda_dist != 0.0) and allmap is not None
        nblk = ((W + 31) // 32) * ((H + 31) // 32)
        dmaps = torch.empty((3, planes, H, W), dtype=torch.float32, device=dev)
        partials = torch.empty((planes * nblk, 2), dtype=torch.float32, device=dev)
        out = torch.empty((6,), dtype=torch.float32, device=dev)
        total = torch.empty((), dtype=torch.float32, device=dev)
        s = _n.current_stream_ptr(dev)
        am = pb = None
        npost = 0
        with torch.cuda.device(dev):
            if reg and planes == 3 and FUSED_LOSS:
                # both halves in one launch (csrc/train_fused.hip): they share no data, their workgroups run side by side
                am = allmap.detach().contiguous().float()
                npost = ((W + 15) // 16) * ((H + 15) // 16)
                pb = torch.empty((npost, 2), dtype=torch.float32, device=dev)
                _check(lib.surfel_train_loss_forward(H, W, _n.ptr(x), _n.ptr(y), _n.ptr(dmaps), _n.ptr(partials), _n.ptr(am), _n.ptr(cam),
                                                     float(depth_ratio), _n.ptr(pb), s), "surfel_train_loss_forward")
            else:
                _check(lib.surfel_l1_ssim_forward(planes, H, W, _n.ptr(x), _n.ptr(y), _n.ptr(dmaps), _n.ptr(partials), s), "surfel_l1_ssim_forward")
                if reg:
                    am = allmap.detach().contiguous().float()
                    npost = ((W + 15) // 16) * ((H + 15) // 16)
                    pb = torch.empty((npost, 2), dtype=torch.float32, device=dev)
                    # maps = NULL: only the two regulariser sums are needed (the backward recomputes from allmap)
                    _check(lib.surfel_render_post_forward(H, W, _n.ptr(am), _n.ptr(cam), float(depth_ratio), None, _n.ptr(pb), s),
                           "surfel_render_post_forward")
            # defer_scalars: the loss scalars are written by an extra workgroup of the fused BACKWARD launch (valid once the backward has
            # run: a training loop reads them after the step) — one launch less per iteration
            ctx.deferred = None
            if defer_scalars and reg and planes == 3 and FUSED_LOSS and (image.requires_grad or getattr(ctx, "manual", False)):
                ctx.deferred = (partials, pb, out, total)
            else:
                _check(lib.surfel_loss_finalize(_n.ptr(partials), planes * nblk, planes * H * W, _n.ptr(pb), npost, H * W, float(lambda_dssim),
                                                float(lambda_normal) if reg else 0.0, float(lambda_dist) if reg else 0.0, _n.ptr(out), _n.ptr(total), s),
                       "surfel_loss_finalize")
        ctx.set_materialize_grads(False)
        ctx.k = (planes, H, W, float(depth_ratio), float(lambda_dssim), float(lambda_normal), float(lambda_dist), reg)
        ctx.shapes = (tuple(image.shape), None if allmap is None else tuple(allmap.shape))
        ctx.save_for_backward(x, y, dmaps, am, cam)
        ctx.mark_non_differentiable(out)
        return total, out

    @staticmethod
    def backward(ctx, g_total, g_out):
        planes, H, W, ratio, lam, ln, ld, reg = ctx.k
        x, y, dmaps, am, cam = ctx.saved_tensors
        if g_total is None:
            return None, None, None, None, None, None, None, None, None
        dev = x.device
        lib = _n.load()
        N = float(planes * H * W)
        g = g_total.contiguous().float().reshape(1)
        grad_img = torch.empty_like(x)
        grad_am = None
        s = _n.current_stream_ptr(dev)
        with torch.cuda.device(dev):
            if reg and planes == 3 and FUSED_LOSS:
                grad_am = torch.empty_like(am)
                fin = ctx.deferred or (None, None, None, None)
                _check(lib.surfel_train_loss_backward(H, W, _n.ptr(x), _n.ptr(y), _n.ptr(dmaps), (1.0 - lam) / N, -lam / N, _n.ptr(am), _n.ptr(cam),
                                                      ratio, ln / (H * W), ld / (H * W), _n.ptr(g), _n.ptr(grad_img), _n.ptr(grad_am),
                                                      _n.ptr(fin[0]), _n.ptr(fin[1]), lam, ln, ld, _n.ptr(fin[2]), _n.ptr(fin[3]), s),
                       "surfel_train_loss_backward")
            else:
                _check(lib.surfel_l1_ssim_backward(planes, H, W, _n.ptr(x), _n.ptr(y), _n.ptr(dmaps), (1.0 - lam) / N, -lam / N, _n.ptr(g), _n.ptr(g),
                                                   _n.ptr(grad_img), s), "surfel_l1_ssim_backward")
                if reg:
                    grad_am = torch.empty_like(am)
                    _check(lib.surfel_render_post_backward(H, W, _n.ptr(am), _n.ptr(cam), ratio, None, ln / (H * W), ld / (H * W), _n.ptr(g),
                                                           _n.ptr(grad_am), s), "surfel_render_post_backward")
        return grad_img.view(ctx.shapes[0]), grad_am, None, None, None, None, None, None, None


def train_loss(image, allmap, gt_image, cam_consts, depth_ratio, lambda_dssim, lambda_normal, lambda_dist, defer_scalars=False):
    """Total loss of train.py:72-88 from the rasterizer's two outputs; cam_consts = the camera's 24-float block
    (surfel_render.post_consts) or None when both regulariser weights are 0.
    defer_scalars: the returned scalars (and the total's VALUE) are filled in by the backward launch instead of a finalize launch
    of their own — for a training loop that calls backward right away and reads them afterwards; gradients do not depend on it."""
    if cam_consts is None:
        if lambda_normal != 0.0 or lambda_dist != 0.0:
            raise ValueError("regularisers need the camera constants")
        cam_consts = torch.empty(0, device=image.device)
        allmap = None
    return _TrainLoss.apply(image, allmap, gt_image, cam_consts, depth_ratio, lambda_dssim, lambda_normal, lambda_dist, defer_scalars)


def train_loss_manual(image, allmap, gt_image, cam_consts, depth_ratio, lambda_dssim, lambda_normal, lambda_dist, defer_scalars=True):
    """train_loss without autograd: returns (ctx, total, scalars); train_loss_manual_backward(ctx, g_total) then yields
    (dL/dimage, dL/dallmap or None).  Same kernels, same bits as the autograd node (call under torch.no_grad())."""
    if cam_consts is None:
        if lambda_normal != 0.0 or lambda_dist != 0.0:
            raise ValueError("regularisers need the camera constants")
        cam_consts = torch.empty(0, device=image.device)
        allmap = None
    ctx = _n.ManualCtx()
    total, out = _TrainLoss.forward(ctx, image, allmap, gt_image, cam_consts, depth_ratio, lambda_dssim, lambda_normal, lambda_dist, defer_scalars)
    return ctx, total, out


def train_loss_manual_backward(ctx, g_total):
    g = _TrainLoss.backward(ctx, g_total, None)
    return g[0], g[1]


class _TrainLossBand(torch.autograd.Function):
    """The training loss of ONE ROW BAND of the image, for tile-band sharded training (surfel_dist, BASELINE config 5).

    Inputs are the band extended by halo rows of the neighbouring bands (surfel_dist.exchange_halo): [3, He, W] image,
    [7, He, W] allmap, the same rows of the target.  The loss kernels run on the extended region as if it were an image; the
    band's share of the full-image sums is read from the per-block partial sums of the blocks that lie inside the band (band and
    halo edges are multiples of the kernels' 32 / 16-row blocks), and the backward scales by the FULL image's pixel count.  Every
    band pixel sits >= 32 rows away from an artificial edge of the extended region, farther than the reach of any loss term it
    takes part in (SSIM: 2 x 5 rows; depth-to-normal stencil: 2 rows), so its gradient equals the unsharded one; gradients that
    land on halo rows are dropped — the rank owning those rows computes them itself.
    Returns (band share of the total loss, sums = [sum |img-gt|, sum SSIM, sum normal error, sum distortion] over the band)."""

    @staticmethod
    def forward(ctx, image, allmap, gt, cam, depth_ratio, lambda_dssim, lambda_normal, lambda_dist, rows, full_hw):
        planes, He, W = _planes(image, gt)
        x = image.detach().contiguous().float(); y = gt.detach().contiguous().float()
        dev = x.device
        lib = _n.load()
        reg = (lambda_normal != 0.0 or lambda_dist != 0.0) and allmap is not None
        a, b = int(rows[0]), int(rows[1])
        if a % 32 != 0 or (b % 32 != 0 and b != He):
            raise ValueError("band rows (%d, %d) of the %d-row extended region must sit on 32-row blocks" % (a, b, He))
        Hf, Wf = int(full_hw[0]), int(full_hw[1])
        nbx, nby = (W + 31) // 32, (He + 31) // 32
        dmaps = torch.empty((3, planes, He, W), dtype=torch.float32, device=dev)
        partials = torch.empty((planes * nbx * nby, 2), dtype=torch.float32, device=dev)
        s = _n.current_stream_ptr(dev)
        am = pb = None
        with torch.cuda.device(dev):
            if reg and planes == 3 and FUSED_LOSS:
                am = allmap.detach().contiguous().float()
                pb = torch.empty((((W + 15) // 16) * ((He + 15) // 16), 2), dtype=torch.float32, device=dev)
                _check(lib.surfel_train_loss_forward(He, W, _n.ptr(x), _n.ptr(y), _n.ptr(dmaps), _n.ptr(partials), _n.ptr(am), _n.ptr(cam),
                                                     float(depth_ratio), _n.ptr(pb), s), "surfel_train_loss_forward")
            else:
                _check(lib.surfel_l1_ssim_forward(planes, He, W, _n.ptr(x), _n.ptr(y), _n.ptr(dmaps), _n.ptr(partials), s), "surfel_l1_ssim_forward")
                if reg:
                    am = allmap.detach().contiguous().float()
                    pb = torch.empty((((W + 15) // 16) * ((He + 15) // 16), 2), dtype=torch.float32, device=dev)
                    _check(lib.surfel_render_post_forward(He, W, _n.ptr(am), _n.ptr(cam), float(depth_ratio), None, _n.ptr(pb), s),
                           "surfel_render_post_forward")
        sums = torch.zeros((4,), dtype=torch.float32, device=dev)
        sums[0:2] = partials.view(planes, nby, nbx, 2)[:, a // 32:(b + 31) // 32].sum((0, 1, 2))
        if reg:
            sums[2:4] = pb.view((He + 15) // 16, (W + 15) // 16, 2)[a // 16:(b + 15) // 16].sum((0, 1))
        N3, N1 = float(planes * Hf * Wf), float(Hf * Wf)
        share = ((1.0 - lambda_dssim) * sums[0] - lambda_dssim * sums[1]) / N3 + (lambda_normal * sums[2] + lambda_dist * sums[3]) / N1
        ctx.set_materialize_grads(False)
        ctx.k = (planes, He, W, float(depth_ratio), float(lambda_dssim), float(lambda_normal), float(lambda_dist), reg, N3, N1)
        ctx.shapes = (tuple(image.shape), None if allmap is None else tuple(allmap.shape))
        ctx.save_for_backward(x, y, dmaps, am, cam)
        ctx.mark_non_differentiable(sums)
        return share, sums

    @staticmethod
    def backward(ctx, g_share, g_sums):
        planes, He, W, ratio, lam, ln, ld, reg, N3, N1 = ctx.k
        x, y, dmaps, am, cam = ctx.saved_tensors
        if g_share is None:
            return (None,) * 10
        dev = x.device
        lib = _n.load()
        g = g_share.contiguous().float().reshape(1)
        grad_img = torch.empty_like(x)
        grad_am = None
        s = _n.current_stream_ptr(dev)
        with torch.cuda.device(dev):
            if reg and planes == 3 and FUSED_LOSS:
                grad_am = torch.empty_like(am)
                _check(lib.surfel_train_loss_backward(He, W, _n.ptr(x), _n.ptr(y), _n.ptr(dmaps), (1.0 - lam) / N3, -lam / N3, _n.ptr(am), _n.ptr(cam),
                                                      ratio, ln / N1, ld / N1, _n.ptr(g), _n.ptr(grad_img), _n.ptr(grad_am),
                                                      None, None, 0.0, 0.0, 0.0, None, None, s),
                       "surfel_train_loss_backward")
            else:
                _check(lib.surfel_l1_ssim_backward(planes, He, W, _n.ptr(x), _n.ptr(y), _n.ptr(dmaps), (1.0 - lam) / N3, -lam / N3, _n.ptr(g), _n.ptr(g),
                                                   _n.ptr(grad_img), s), "surfel_l1_ssim_backward")
                if reg:
                    grad_am = torch.empty_like(am)
                    _check(lib.surfel_render_post_backward(He, W, _n.ptr(am), _n.ptr(cam), ratio, None, ln / N1, ld / N1, _n.ptr(g),
                                                           _n.ptr(grad_am), s), "surfel_render_post_backward")
        return (grad_img.view(ctx.shapes[0]), grad_am) + (None,) * 8


def train_loss_band(image_ext, allmap_ext, gt_ext, cam_consts_ext, depth_ratio, lambda_dssim, lambda_normal, lambda_dist, rows, full_hw):
    """Band share of train.py:72-88's loss for tile-band sharding: the shares of all ranks add up to the full-image loss
    (without the constant lambda_dssim term), their gradients are the unsharded ones.  rows = (first, end) band rows inside the
    extended region, full_hw = (H, W) of the whole image, cam_consts_ext = surfel_render.post_consts_rows(...)."""
    if cam_consts_ext is None:
        if lambda_normal != 0.0 or lambda_dist != 0.0:
            raise ValueError("regularisers need the camera constants")
        cam_consts_ext = torch.empty(0, device=image_ext.device)
        allmap_ext = None
    return _TrainLossBand.apply(image_ext, allmap_ext, gt_ext, cam_consts_ext, depth_ratio, lambda_dssim, lambda_normal, lambda_dist, rows, full_hw)


def scalars_from_band_sums(sums, planes_hw, hw, lambda_dssim, lambda_normal, lambda_dist):
    """[Ll1, ssim, normal_err, dist, photometric, total] (the layout of train_loss's scalars) from the all-reduced band sums."""
    l1, ss, ne, di = sums[0] / planes_hw, sums[1] / planes_hw, sums[2] / hw, sums[3] / hw
    photo = (1.0 - lambda_dssim) * l1 + lambda_dssim * (1.0 - ss)
    return torch.stack([l1, ss, ne, di, photo, photo + lambda_normal * ne + lambda_dist * di])
