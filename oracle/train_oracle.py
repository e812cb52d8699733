"""TEST INFRASTRUCTURE ONLY — CPU restatement (torch fp64 + autograd) of the reference's training-side maths around the
rasterizer.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this; the product path (libsurfel_hip.so through surfel_train.py) never does.

Pinned against the reference itself: tests/golden/ref_train.npz is produced by tests/golden/make_golden_train.py, which
imports /root/reference's own loss_utils / render() / GaussianModel; tests/test_train_oracle_cpu.py checks every function
here against those vectors.  Each function cites the reference lines it follows.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

DT = torch.float64


def gaussian_window(window_size=11, sigma=1.5, dtype=DT):
    """utils/loss_utils.py:29-31, 42-46: normalised 1-D gaussian, outer product -> [1,1,11,11]."""
    g = torch.tensor([math.exp(-(x - window_size // 2) ** 2 / float(2 * sigma ** 2)) for x in range(window_size)], dtype=dtype)
    g = g / g.sum()
    return (g[:, None] @ g[None, :])[None, None]


def l1_loss(x, y):
    """utils/loss_utils.py:23-24."""
    return (x - y).abs().mean()


def ssim(img1, img2, window_size=11):
    """utils/loss_utils.py:43-73 (size_average=True): grouped 11x11 gaussian correlation with zero padding."""
    C = img1.shape[-3]
    w = gaussian_window(window_size, 1.5, img1.dtype).expand(C, 1, window_size, window_size).contiguous()
    pad = window_size // 2
    a = img1[None] if img1.dim() == 3 else img1
    b = img2[None] if img2.dim() == 3 else img2
    mu1 = F.conv2d(a, w, padding=pad, groups=C); mu2 = F.conv2d(b, w, padding=pad, groups=C)
    mu1_sq, mu2_sq, mu12 = mu1 * mu1, mu2 * mu2, mu1 * mu2
    s1 = F.conv2d(a * a, w, padding=pad, groups=C) - mu1_sq
    s2 = F.conv2d(b * b, w, padding=pad, groups=C) - mu2_sq
    s12 = F.conv2d(a * b, w, padding=pad, groups=C) - mu12
    C1, C2 = 0.01 ** 2, 0.03 ** 2
    m = ((2 * mu12 + C1) * (2 * s12 + C2)) / ((mu1_sq + mu2_sq + C1) * (s1 + s2 + C2))
    return m.mean()


def photometric(img, gt, lambda_dssim=0.2):
    """train.py:72-74.  Returns dict of values and gradients w.r.t. img (numpy fp64)."""
    x = torch.tensor(np.asarray(img), dtype=DT, requires_grad=True); y = torch.tensor(np.asarray(gt), dtype=DT)
    Ll1 = l1_loss(x, y); s = ssim(x, y)
    loss = (1.0 - lambda_dssim) * Ll1 + lambda_dssim * (1.0 - s)
    g_l1, = torch.autograd.grad(Ll1, x, retain_graph=True)
    g_ss, = torch.autograd.grad(s, x, retain_graph=True)
    g, = torch.autograd.grad(loss, x)
    return dict(l1=Ll1.item(), ssim=s.item(), loss=loss.item(), g_l1=g_l1.numpy(), g_ssim=g_ss.numpy(), g_loss=g.numpy())


def depths_to_points(wvt, fpt, W, H, depthmap):
    """utils/point_utils.py:9-24 (wvt = world_view_transform, fpt = full_proj_transform, both as the Camera stores them)."""
    c2w = torch.linalg.inv(wvt.T)
    ndc2pix = torch.tensor([[W / 2, 0, 0, W / 2], [0, H / 2, 0, H / 2], [0, 0, 0, 1]], dtype=wvt.dtype).T
    projection_matrix = c2w.T @ fpt
    intrins = (projection_matrix @ ndc2pix)[:3, :3].T
    gx, gy = torch.meshgrid(torch.arange(W, dtype=wvt.dtype), torch.arange(H, dtype=wvt.dtype), indexing="xy")
    pts = torch.stack([gx, gy, torch.ones_like(gx)], dim=-1).reshape(-1, 3)
    rays_d = pts @ torch.linalg.inv(intrins).T @ c2w[:3, :3].T
    rays_o = c2w[:3, 3]
    return depthmap.reshape(-1, 1) * rays_d + rays_o


def depth_to_normal(wvt, fpt, W, H, depth):
    """utils/point_utils.py:26-37: central differences of the back-projected points, zero on the 1-pixel border."""
    points = depths_to_points(wvt, fpt, W, H, depth).reshape(H, W, 3)
    out = torch.zeros_like(points)
    dx = points[2:, 1:-1] - points[:-2, 1:-1]
    dy = points[1:-1, 2:] - points[1:-1, :-2]
    out[1:-1, 1:-1, :] = F.normalize(torch.cross(dx, dy, dim=-1), dim=-1)
    return out


def cam_consts(wvt, fpt, W, H):
    """The 24-float camera block of include/surfel_train.h, derived with the reference's own formulas (fp64)."""
    wvt = torch.tensor(np.asarray(wvt), dtype=DT); fpt = torch.tensor(np.asarray(fpt), dtype=DT)
    c2w = torch.linalg.inv(wvt.T)
    ndc2pix = torch.tensor([[W / 2, 0, 0, W / 2], [0, H / 2, 0, H / 2], [0, 0, 0, 1]], dtype=DT).T
    intrins = ((c2w.T @ fpt) @ ndc2pix)[:3, :3].T
    K = torch.linalg.inv(intrins).T @ c2w[:3, :3].T
    out = np.zeros(24, np.float64)
    out[0:9] = wvt[:3, :3].numpy().reshape(-1)
    out[9:18] = K.numpy().reshape(-1)
    out[18:21] = c2w[:3, 3].numpy()
    return out


def render_post(allmap, wvt, fpt, W, H, depth_ratio):
    """gaussian_renderer/__init__.py:118-147 on a torch allmap [7,H,W]; returns maps [9,H,W]:
    0 rend_alpha | 1-3 rend_normal | 4 rend_dist | 5 surf_depth | 6-8 surf_normal."""
    render_alpha = allmap[1:2]
    render_normal = (allmap[2:5].permute(1, 2, 0) @ (wvt[:3, :3].T)).permute(2, 0, 1)
    med = torch.nan_to_num(allmap[5:6], 0, 0)
    exp = torch.nan_to_num(allmap[0:1] / render_alpha, 0, 0)
    surf_depth = exp * (1 - depth_ratio) + depth_ratio * med
    surf_normal = depth_to_normal(wvt, fpt, W, H, surf_depth).permute(2, 0, 1) * render_alpha.detach()
    return torch.cat([render_alpha, render_normal, allmap[6:7], surf_depth, surf_normal], dim=0)


def render_post_np(allmap, wvt, fpt, W, H, depth_ratio, wmaps=None, lambda_normal=0.0, lambda_dist=0.0):
    """Values of the maps, the regulariser means (train.py:80-85) and gradients w.r.t. allmap of
    sum(maps * wmaps) and of lambda_normal * mean(1 - rn.sn) + lambda_dist * mean(dist).  NaN gradients (the 0/0 pixels of
    the reference's autograd) are returned as NaN; callers treat them as don't-care."""
    am = torch.tensor(np.asarray(allmap), dtype=DT, requires_grad=True)
    wvt = torch.tensor(np.asarray(wvt), dtype=DT); fpt = torch.tensor(np.asarray(fpt), dtype=DT)
    maps = render_post(am, wvt, fpt, W, H, depth_ratio)
    normal_err = (1 - (maps[1:4] * maps[6:9]).sum(dim=0)).mean()
    dist_mean = maps[4].mean()
    out = dict(maps=maps.detach().numpy(), normal_err_mean=normal_err.item(), dist_mean=dist_mean.item())
    if wmaps is not None:
        out["g_maps"], = (g.numpy() for g in torch.autograd.grad((maps * torch.tensor(np.asarray(wmaps), dtype=DT)).sum(), am, retain_graph=True))
    if lambda_normal or lambda_dist:
        out["g_reg"], = (g.numpy() for g in torch.autograd.grad(lambda_normal * normal_err + lambda_dist * dist_mean, am))
    return out


# ------------------------------------------------------------------------------------------------ parameter store
SECTIONS = (("xyz", 3), ("opacity", 1), ("scaling", 2), ("rotation", 4), ("sh", 48))


def activate(opacity, scaling, rotation):
    """scene/gaussian_model.py:95-115: sigmoid / exp / F.normalize."""
    o = torch.sigmoid(torch.as_tensor(opacity, dtype=DT)); s = torch.exp(torch.as_tensor(scaling, dtype=DT))
    r = F.normalize(torch.as_tensor(rotation, dtype=DT))
    return o.numpy(), s.numpy(), r.numpy()


class AdamOracle:
    """torch.optim.Adam(lr=0, eps=1e-15) over the six groups of scene/gaussian_model.py:153-162, fp64, driven by
    gradients w.r.t. the ACTIVATED values exactly like the reference's autograd chain (get_* properties)."""

    def __init__(self, xyz, f_dc, f_rest, opacity, scaling, rotation, eps=1e-15):
        mk = lambda a: torch.nn.Parameter(torch.tensor(np.asarray(a), dtype=DT))
        self.p = dict(xyz=mk(xyz), f_dc=mk(f_dc), f_rest=mk(f_rest), opacity=mk(opacity), scaling=mk(scaling), rotation=mk(rotation))
        self.names = ["xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation"]
        self.opt = torch.optim.Adam([{"params": [self.p[n]], "lr": 0.0, "name": n} for n in self.names], lr=0.0, eps=eps)

    def step(self, lrs, g_xyz, g_features, g_opacity, g_scaling, g_rotation):
        for grp, lr in zip(self.opt.param_groups, lrs):
            grp["lr"] = float(lr)
        t = lambda a: torch.tensor(np.asarray(a), dtype=DT)
        p = self.p
        feats = torch.cat((p["f_dc"], p["f_rest"]), dim=1)
        sur = (p["xyz"] * t(g_xyz)).sum() + (feats * t(g_features)).sum() + (torch.sigmoid(p["opacity"]) * t(g_opacity)).sum() + \
            (torch.exp(p["scaling"]) * t(g_scaling)).sum() + (F.normalize(p["rotation"]) * t(g_rotation)).sum()
        sur.backward()
        self.opt.step(); self.opt.zero_grad(set_to_none=True)
        return {n: v.detach().numpy().copy() for n, v in p.items()}


def expon_lr(step, lr_init, lr_final, lr_delay_steps=0, lr_delay_mult=1.0, max_steps=1000000):
    """utils/general_utils.py:39-70 (get_expon_lr_func): log-linear interpolation with optional delay."""
    if step < 0 or (lr_init == 0.0 and lr_final == 0.0):
        return 0.0
    if lr_delay_steps > 0:
        delay_rate = lr_delay_mult + (1 - lr_delay_mult) * np.sin(0.5 * np.pi * np.clip(step / lr_delay_steps, 0, 1))
    else:
        delay_rate = 1.0
    t = np.clip(step / max_steps, 0, 1)
    return delay_rate * np.exp(np.log(lr_init) * (1 - t) + np.log(lr_final) * t)


def densify_stats(accum, denom, max_radii, g2d, radii):
    """train.py:127 + scene/gaussian_model.py:405-407 for one view (numpy, in place on copies)."""
    accum, denom, max_radii = accum.copy(), denom.copy(), max_radii.copy()
    vis = radii > 0
    max_radii[vis] = np.maximum(max_radii[vis], radii[vis].astype(max_radii.dtype))
    accum[vis] += np.linalg.norm(g2d[vis].astype(np.float64), axis=-1)
    denom[vis] += 1
    return accum, denom, max_radii
