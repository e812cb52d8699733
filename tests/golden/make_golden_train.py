"""Generate tests/golden/ref_train.npz by IMPORTING the reference's own Python code for the training-side rows
(SURVEY.md §8f N1-N3).  Runs only in the build container (needs /root/reference); the .npz is committed.

Pinned from the reference itself (all of this IS in-tree, unlike the rasterizer):
  * l1_loss, ssim                              utils/loss_utils.py:23-24, 43-73   (values + autograd gradients)
  * render()'s allmap post-processing          gaussian_renderer/__init__.py:118-147 with utils/point_utils.py:9-37,
    for depth_ratio 0 and 1, driven through the reference's real render() by a stub rasterizer that returns a
    prescribed allmap; values of the five maps, the regularisers of train.py:80-85 and d(loss)/d(allmap) by autograd
  * GaussianModel activations + optimiser      scene/gaussian_model.py:95-115, 148-166: three optimizer.step()s of the
    reference's own Adam setup on the reference's own parameter groups, driven by prescribed gradients w.r.t. the
    ACTIVATED values (chained through the reference's activations by autograd)
  * densification statistics                   scene/gaussian_model.py:405-407, train.py:127

Usage:  python tests/golden/make_golden_train.py
"""
import math
import os
import sys
import types
from typing import NamedTuple

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REF = "/root/reference"
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "2d-gaussian-splatting_amd"))

for name in ["plyfile", "cv2", "matplotlib", "matplotlib.pyplot", "simple_knn", "simple_knn._C", "diff_surfel_rasterization"]:
    sys.modules[name] = types.ModuleType(name)
sys.modules["plyfile"].PlyData = object
sys.modules["plyfile"].PlyElement = object
sys.modules["simple_knn._C"].distCUDA2 = lambda x: None
sys.modules["matplotlib"].pyplot = sys.modules["matplotlib.pyplot"]
torch.Tensor.cuda = lambda self, *a, **k: self


def _cpu_factory(fn):
    def wrapped(*a, **k):
        if "device" in k and str(k["device"]).startswith("cuda"):
            k["device"] = "cpu"
        return fn(*a, **k)
    return wrapped


for _n in ["zeros", "ones", "tensor", "arange", "empty", "zeros_like", "ones_like", "rand", "randn", "full"]:
    setattr(torch, _n, _cpu_factory(getattr(torch, _n)))

NEXT_ALLMAP = {}


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


class GaussianRasterizer:
    """Stub: hands render() a prescribed (color, radii, allmap) so that its post-processing runs on known inputs."""

    def __init__(self, raster_settings):
        self.rs = raster_settings

    def __call__(self, means3D, means2D, shs=None, colors_precomp=None, opacities=None, scales=None, rotations=None,
                 cov3D_precomp=None):
        return NEXT_ALLMAP["color"], NEXT_ALLMAP["radii"], NEXT_ALLMAP["allmap"]


sys.modules["diff_surfel_rasterization"].GaussianRasterizationSettings = GaussianRasterizationSettings
sys.modules["diff_surfel_rasterization"].GaussianRasterizer = GaussianRasterizer

sys.path.insert(0, REF)
from gaussian_renderer import render                      # noqa: E402
from scene.cameras import Camera                          # noqa: E402
from scene.gaussian_model import GaussianModel            # noqa: E402
from utils.loss_utils import l1_loss, ssim                # noqa: E402
from arguments import OptimizationParams                  # noqa: E402

import synthetic                                          # noqa: E402


def make_allmap(rng, H, W):
    """A plausible rasterizer output: smooth depth field with holes (alpha exactly 0 -> 0/0), un-normalised normals."""
    yy, xx = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
    depth = 3.0 + 0.8 * np.sin(xx / 7.0) + 0.5 * np.cos(yy / 5.0) + 0.05 * rng.normal(size=(H, W))
    alpha = np.clip(0.6 + 0.5 * np.sin(xx / 11.0 + yy / 13.0) + 0.1 * rng.normal(size=(H, W)), 0.0, 0.999)
    hole = (xx - W * 0.3) ** 2 + (yy - H * 0.6) ** 2 < 36
    alpha[hole] = 0.0
    am = np.zeros((7, H, W), np.float32)
    am[0] = depth * alpha
    am[1] = alpha
    n = rng.normal(size=(3, H, W)); n /= np.linalg.norm(n, axis=0, keepdims=True)
    am[2:5] = n * alpha
    am[5] = np.where(alpha > 0.5, depth + 0.02 * rng.normal(size=(H, W)), 0.0)
    am[6] = np.abs(rng.normal(size=(H, W))) * 0.01 * alpha
    return am.astype(np.float32)


def main():
    out = {}
    rng = np.random.default_rng(5)
    # ------------------------------------------------------------------ losses
    for tag, (C, H, W) in {"a": (3, 45, 70), "b": (3, 64, 64), "c": (1, 9, 7)}.items():
        gt = rng.uniform(0, 1, size=(C, H, W)).astype(np.float32)
        img = np.clip(gt + 0.15 * rng.normal(size=(C, H, W)), -0.1, 1.2).astype(np.float32)
        if tag == "a":
            img[:, :5, :5] = gt[:, :5, :5]          # exact ties: sign(0) = 0 in the L1 gradient
        x = torch.tensor(img, requires_grad=True); y = torch.tensor(gt)
        Ll1 = l1_loss(x, y); s = ssim(x, y)
        loss = 0.8 * Ll1 + 0.2 * (1.0 - s)          # train.py:73-74 with lambda_dssim = 0.2
        g_l1, = torch.autograd.grad(Ll1, x, retain_graph=True)
        g_ss, = torch.autograd.grad(s, x, retain_graph=True)
        g_loss, = torch.autograd.grad(loss, x)
        out.update({"loss_%s_img" % tag: img, "loss_%s_gt" % tag: gt, "loss_%s_l1" % tag: Ll1.item(), "loss_%s_ssim" % tag: s.item(),
                    "loss_%s_total" % tag: loss.item(), "loss_%s_g_l1" % tag: g_l1.numpy(), "loss_%s_g_ssim" % tag: g_ss.numpy(),
                    "loss_%s_g_total" % tag: g_loss.numpy()})

    # ------------------------------------------------------------------ render() post-processing
    W, H = 72, 56
    sc = synthetic.make_scene(16, W, H, seed=3, view_index=2)     # a rotated, translated camera
    Rt = sc["viewmatrix"].T.astype(np.float64)
    cam = Camera(colmap_id=0, R=Rt[:3, :3].T, T=Rt[:3, 3], FoVx=2 * math.atan(sc["tanfovx"]), FoVy=2 * math.atan(sc["tanfovy"]),
                 image=torch.zeros(3, H, W), gt_alpha_mask=None, image_name="synthetic", uid=0)
    pc = GaussianModel(3)
    pc._xyz = torch.zeros(16, 3); pc._scaling = torch.zeros(16, 2); pc._rotation = torch.ones(16, 4); pc._opacity = torch.zeros(16, 1)
    pc._features_dc = torch.zeros(16, 1, 3); pc._features_rest = torch.zeros(16, 15, 3)
    am_np = make_allmap(rng, H, W)
    wmaps = rng.normal(size=(9, H, W)).astype(np.float32)       # upstream gradient of the 9 map channels
    out.update(post_W=W, post_H=H, post_world_view_transform=cam.world_view_transform.numpy(),
               post_full_proj_transform=cam.full_proj_transform.numpy(), post_allmap=am_np, post_wmaps=wmaps)
    for ratio in (0.0, 1.0):
        pipe = types.SimpleNamespace(compute_cov3D_python=False, convert_SHs_python=False, depth_ratio=ratio, debug=False)
        am = torch.tensor(am_np, requires_grad=True)
        NEXT_ALLMAP.update(color=torch.zeros(3, H, W), radii=torch.ones(16, dtype=torch.int32), allmap=am)
        r = render(cam, pc, pipe, torch.zeros(3))
        maps = torch.cat([r["rend_alpha"], r["rend_normal"], r["rend_dist"], r["surf_depth"], r["surf_normal"]], dim=0)
        # train.py:80-85
        normal_error = (1 - (r["rend_normal"] * r["surf_normal"]).sum(dim=0))[None]
        lam_n, lam_d = 0.05, 100.0
        normal_loss = lam_n * normal_error.mean(); dist_loss = lam_d * r["rend_dist"].mean()
        g_reg, = torch.autograd.grad(normal_loss + dist_loss, am, retain_graph=True)
        g_maps, = torch.autograd.grad((maps * torch.tensor(wmaps)).sum(), am)
        t = "post_r%d" % int(ratio)
        out.update({t + "_maps": maps.detach().numpy(), t + "_normal_err_mean": normal_error.mean().item(),
                    t + "_dist_mean": r["rend_dist"].mean().item(), t + "_g_reg": g_reg.numpy(), t + "_g_maps": g_maps.numpy()})
    out.update(post_lambda_normal=0.05, post_lambda_dist=100.0)

    # ------------------------------------------------------------------ GaussianModel: activations + the reference's Adam
    P = 40
    import argparse
    opt = OptimizationParams(argparse.ArgumentParser())
    gm = GaussianModel(3)
    gm.spatial_lr_scale = 2.5
    gm._xyz = torch.nn.Parameter(torch.tensor(rng.normal(size=(P, 3)).astype(np.float32)))
    gm._features_dc = torch.nn.Parameter(torch.tensor(rng.normal(size=(P, 1, 3)).astype(np.float32)))
    gm._features_rest = torch.nn.Parameter(torch.tensor(0.1 * rng.normal(size=(P, 15, 3)).astype(np.float32)))
    gm._scaling = torch.nn.Parameter(torch.tensor(rng.normal(-2.0, 0.5, size=(P, 2)).astype(np.float32)))
    gm._rotation = torch.nn.Parameter(torch.tensor(rng.normal(size=(P, 4)).astype(np.float32) * np.linspace(0.3, 3, P)[:, None].astype(np.float32)))
    gm._opacity = torch.nn.Parameter(torch.tensor(rng.normal(0, 2, size=(P, 1)).astype(np.float32)))
    gm.max_radii2D = torch.zeros(P)
    gm.training_setup(opt)
    out.update(adam_P=P, adam_spatial_lr_scale=2.5,
               adam_theta0_xyz=gm._xyz.detach().numpy().copy(), adam_theta0_f_dc=gm._features_dc.detach().numpy().copy(),
               adam_theta0_f_rest=gm._features_rest.detach().numpy().copy(), adam_theta0_opacity=gm._opacity.detach().numpy().copy(),
               adam_theta0_scaling=gm._scaling.detach().numpy().copy(), adam_theta0_rotation=gm._rotation.detach().numpy().copy(),
               act0_opacity=gm.get_opacity.detach().numpy(), act0_scaling=gm.get_scaling.detach().numpy(),
               act0_rotation=gm.get_rotation.detach().numpy(), act0_features=gm.get_features.detach().numpy())
    lrs = []
    for it in (1, 2, 3):
        lr_xyz = gm.update_learning_rate(it)
        lrs.append([g["lr"] for g in gm.optimizer.param_groups])
        # gradients w.r.t. what the rasterizer consumes (activated values), prescribed
        g_xyz = torch.tensor(rng.normal(size=(P, 3)).astype(np.float32)) * 1e-3
        g_feat = torch.tensor(rng.normal(size=(P, 16, 3)).astype(np.float32)) * 1e-3
        g_op = torch.tensor(rng.normal(size=(P, 1)).astype(np.float32)) * 1e-2
        g_sc = torch.tensor(rng.normal(size=(P, 2)).astype(np.float32)) * 1e-2
        g_rot = torch.tensor(rng.normal(size=(P, 4)).astype(np.float32)) * 1e-3
        g_rot[:5] = 0.0                                           # zero gradients still move by momentum
        surrogate = (gm.get_xyz * g_xyz).sum() + (gm.get_features * g_feat).sum() + (gm.get_opacity * g_op).sum() + \
            (gm.get_scaling * g_sc).sum() + (gm.get_rotation * g_rot).sum()
        surrogate.backward()
        gm.optimizer.step()
        gm.optimizer.zero_grad(set_to_none=True)
        if it == 2:        # the reference's own checkpoint tuple (train.py:142-144) after two iterations
            torch.save((gm.capture(), it), os.path.join(os.path.dirname(os.path.abspath(__file__)), "ref_chkpnt2.pth"))
        out.update({"adam_g%d_xyz" % it: g_xyz.numpy(), "adam_g%d_features" % it: g_feat.numpy(), "adam_g%d_opacity" % it: g_op.numpy(),
                    "adam_g%d_scaling" % it: g_sc.numpy(), "adam_g%d_rotation" % it: g_rot.numpy(),
                    "adam_theta%d_xyz" % it: gm._xyz.detach().numpy().copy(), "adam_theta%d_f_dc" % it: gm._features_dc.detach().numpy().copy(),
                    "adam_theta%d_f_rest" % it: gm._features_rest.detach().numpy().copy(),
                    "adam_theta%d_opacity" % it: gm._opacity.detach().numpy().copy(),
                    "adam_theta%d_scaling" % it: gm._scaling.detach().numpy().copy(),
                    "adam_theta%d_rotation" % it: gm._rotation.detach().numpy().copy()})
    out["adam_lrs"] = np.array(lrs, np.float64)       # rows: iteration 1..3; columns: xyz, f_dc, f_rest, opacity, scaling, rotation
    out["adam_eps"] = gm.optimizer.defaults["eps"]; out["adam_betas"] = np.array(gm.optimizer.defaults["betas"])
    out["xyz_lr_at"] = np.array([[it, gm.xyz_scheduler_args(it)] for it in (1, 100, 1000, 7000, 15000, 30000)], np.float64)

    # ------------------------------------------------------------------ densification statistics (two views)
    vs = torch.zeros(P, 3, requires_grad=True)
    for view in range(2):
        g2d = torch.tensor(rng.normal(size=(P, 3)).astype(np.float32)); g2d[:, 2] = 0
        radii = torch.tensor(rng.integers(0, 30, size=P).astype(np.int32)); radii[::4] = 0
        vs.grad = g2d
        vis = radii > 0
        gm.max_radii2D[vis] = torch.max(gm.max_radii2D[vis], radii[vis])      # train.py:127
        gm.add_densification_stats(vs, vis)                                    # train.py:128
        out.update({"dens_g2d_%d" % view: g2d.numpy(), "dens_radii_%d" % view: radii.numpy()})
    out.update(dens_accum=gm.xyz_gradient_accum.numpy(), dens_denom=gm.denom.numpy(), dens_max_radii=gm.max_radii2D.numpy())

    # ------------------------------------------------------------------ densify_and_prune (scene/gaussian_model.py:348-403)
    P = 300
    dm = GaussianModel(3)
    dm.spatial_lr_scale = 1.0
    dm._xyz = torch.nn.Parameter(torch.tensor(rng.normal(size=(P, 3)).astype(np.float32)))
    dm._features_dc = torch.nn.Parameter(torch.tensor(rng.normal(size=(P, 1, 3)).astype(np.float32)))
    dm._features_rest = torch.nn.Parameter(torch.tensor(0.1 * rng.normal(size=(P, 15, 3)).astype(np.float32)))
    dm._scaling = torch.nn.Parameter(torch.tensor(rng.normal(-3.0, 0.8, size=(P, 2)).astype(np.float32)))
    dm._rotation = torch.nn.Parameter(torch.tensor(rng.normal(size=(P, 4)).astype(np.float32)))
    dm._opacity = torch.nn.Parameter(torch.tensor(rng.normal(-1.0, 2.0, size=(P, 1)).astype(np.float32)))
    dm.max_radii2D = torch.tensor(rng.integers(0, 40, size=P).astype(np.float32))
    dm.training_setup(opt)
    # give the optimiser a state to carry through the re-indexing
    (dm.get_xyz.sum() + dm.get_features.sum() + dm.get_opacity.sum() + dm.get_scaling.sum() + dm.get_rotation.sum()).backward()
    dm.update_learning_rate(1); dm.optimizer.step(); dm.optimizer.zero_grad(set_to_none=True)
    dm.xyz_gradient_accum = torch.tensor(np.abs(rng.normal(0, 4e-4, size=(P, 1))).astype(np.float32))
    dm.denom = torch.tensor(rng.integers(0, 3, size=(P, 1)).astype(np.float32))          # zeros -> NaN grads -> 0
    extent, max_grad, min_opacity, max_screen = 4.0, 0.0002, 0.05, 20
    out.update(dens2_in_accum=dm.xyz_gradient_accum.numpy().copy(), dens2_in_denom=dm.denom.numpy().copy(),
               dens2_in_max_radii=dm.max_radii2D.numpy().copy())
    before = {k: getattr(dm, "_" + k).detach().numpy().copy() for k in ("xyz", "features_dc", "features_rest", "scaling", "rotation", "opacity")}
    before_m = dm.optimizer.state[dm.optimizer.param_groups[0]["params"][0]]["exp_avg"].numpy().copy()
    torch.manual_seed(0)
    dm.densify_and_prune(max_grad, min_opacity, extent, max_screen)
    out.update(dens2_extent=extent, dens2_max_grad=max_grad, dens2_min_opacity=min_opacity, dens2_max_screen=max_screen,
               dens2_P_after=dm.get_xyz.shape[0], dens2_percent_dense=opt.percent_dense)
    for k, v in before.items():
        out["dens2_before_" + k] = v
    out["dens2_before_m_xyz"] = before_m
    for k in ("xyz", "features_dc", "features_rest", "scaling", "rotation", "opacity"):
        out["dens2_after_" + k] = getattr(dm, "_" + k).detach().numpy().copy()
    out["dens2_after_m_xyz"] = dm.optimizer.state[dm.optimizer.param_groups[0]["params"][0]]["exp_avg"].numpy().copy()
    out["dens2_after_accum_sum"] = float(dm.xyz_gradient_accum.sum()); out["dens2_after_maxr_sum"] = float(dm.max_radii2D.sum())

    # ------------------------------------------------------------------ quaternion -> rotation, splat2world (utils/general_utils.py:78-110)
    from utils.general_utils import build_rotation
    q = torch.tensor(rng.normal(size=(64, 4)).astype(np.float32))
    out.update(rot_q=q.numpy(), rot_R=build_rotation(q).numpy())
    cm = GaussianModel(3)
    cm._xyz = torch.tensor(rng.normal(size=(64, 3)).astype(np.float32)); cm._scaling = torch.tensor(rng.normal(-2, 0.5, size=(64, 2)).astype(np.float32))
    cm._rotation = q
    out.update(cov_xyz=cm._xyz.numpy(), cov_scaling=cm._scaling.numpy(), cov_splat2world=cm.get_covariance(1.7).numpy())

    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ref_train.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes,", len(out), "arrays")


if __name__ == "__main__":
    main()
