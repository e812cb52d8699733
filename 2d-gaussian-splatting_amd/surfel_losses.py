"""Photometric losses of the reference's training loop on MI355X — drop-in for `utils.loss_utils.l1_loss` / `ssim`
(/root/reference/utils/loss_utils.py:23-24, 43-73; used at train.py:72-74) plus the fused form the trainer uses.

Same names, argument meaning and return shapes; the work is done by two LDS-tiled HIP kernels of libsurfel_hip.so
(include/surfel_train.h) instead of five grouped conv2d calls + ~40 elementwise kernels per iteration.
No CPU / PyTorch fallback: CPU tensors raise.
"""
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
    def forward(ctx, img, gt):
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
            _check(lib.surfel_l1_ssim_forward(planes, H, W, _n.ptr(x), _n.ptr(y), _n.ptr(dmaps), _n.ptr(partials), s), "surfel_l1_ssim_forward")
            _check(lib.surfel_reduce_partials(_n.ptr(partials), 1, planes * nblk, 2, 1.0 / (planes * H * W), _n.ptr(out), s),
                   "surfel_reduce_partials")
        ctx.dims = (planes, H, W)
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
            _check(lib.surfel_l1_ssim_backward(planes, H, W, _n.ptr(x), _n.ptr(y), _n.ptr(dmaps), 1.0 / N, 1.0 / N, _n.ptr(g[0:1]), _n.ptr(g[1:2]),
                                               _n.ptr(grad), _n.current_stream_ptr(dev)), "surfel_l1_ssim_backward")
        return grad.view(ctx.in_shape), None


def _means(img, gt):
    out = _L1SSIM.apply(img, gt)
    return out


def l1_loss(network_output, gt):
    """mean |network_output - gt| (utils/loss_utils.py:23-24)."""
    return _means(network_output, gt)[0]


def ssim(img1, img2, window_size=11, size_average=True):
    """Mean SSIM with the reference's 11x11 sigma-1.5 window and zero padding (utils/loss_utils.py:43-73)."""
    if window_size != 11:
        raise NotImplementedError("the HIP kernel implements the reference's window_size=11 only")
    if not size_average:
        raise NotImplementedError("size_average=False is not used by the reference's training loop")
    return _means(img1, img2)[1]


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
