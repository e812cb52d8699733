"""CPU: pin oracle/train_oracle.py against tests/golden/ref_train.npz — vectors produced by the reference's OWN Python
(loss_utils, render() post-processing, GaussianModel + its Adam setup; tests/golden/make_golden_train.py)."""
import os

import numpy as np
import pytest

from oracle import train_oracle as T

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def gt():
    return np.load(os.path.join(REPO, "tests", "golden", "ref_train.npz"))


def close(a, b, atol, rtol):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return np.abs(a - b) <= atol + rtol * np.abs(b)


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_l1_ssim_match_reference(gt, tag):
    r = T.photometric(gt["loss_%s_img" % tag], gt["loss_%s_gt" % tag], 0.2)
    # reference ran in fp32, the oracle in fp64
    assert abs(r["l1"] - gt["loss_%s_l1" % tag]) < 2e-6
    assert abs(r["ssim"] - gt["loss_%s_ssim" % tag]) < 2e-5
    assert abs(r["loss"] - gt["loss_%s_total" % tag]) < 2e-5
    n = gt["loss_%s_img" % tag].size
    assert close(r["g_l1"], gt["loss_%s_g_l1" % tag], 1e-9, 1e-5).all()
    assert close(r["g_ssim"], gt["loss_%s_g_ssim" % tag], 2e-4 / n, 2e-3).mean() > 0.999
    assert close(r["g_loss"], gt["loss_%s_g_total" % tag], 2e-4 / n, 2e-3).mean() > 0.999


@pytest.mark.parametrize("ratio", [0, 1])
def test_render_post_matches_reference_render(gt, ratio):
    W, H = int(gt["post_W"]), int(gt["post_H"])
    r = T.render_post_np(gt["post_allmap"], gt["post_world_view_transform"], gt["post_full_proj_transform"], W, H, float(ratio),
                         wmaps=gt["post_wmaps"], lambda_normal=float(gt["post_lambda_normal"]), lambda_dist=float(gt["post_lambda_dist"]))
    t = "post_r%d" % ratio
    ref = gt[t + "_maps"]
    assert close(r["maps"][:6], ref[:6], 1e-5, 1e-5).all()
    # finite-difference normals amplify fp32 rounding of the reference's points where the surface is nearly degenerate
    assert close(r["maps"][6:], ref[6:], 2e-3, 1e-3).mean() > 0.995
    assert abs(r["normal_err_mean"] - gt[t + "_normal_err_mean"]) < 1e-4
    assert abs(r["dist_mean"] - gt[t + "_dist_mean"]) < 1e-7
    for k in ("g_maps", "g_reg"):
        a, b = r[k], gt[t + "_" + k]
        ok = np.isfinite(b)          # 0/0 pixels: the reference's autograd yields NaN (never read by the rasterizer's backward)
        assert (np.isfinite(a) == ok).all()
        scale = np.abs(b[ok]).mean()
        assert close(a[ok], b[ok], 2e-3 * scale, 2e-2).mean() > 0.99, k
        cos = (a[ok] * b[ok]).sum() / np.sqrt((a[ok] ** 2).sum() * (b[ok] ** 2).sum())
        assert cos > 0.9999, (k, cos)


def test_activations_and_adam_match_reference_model(gt):
    o, s, r = T.activate(gt["adam_theta0_opacity"], gt["adam_theta0_scaling"], gt["adam_theta0_rotation"])
    assert close(o, gt["act0_opacity"], 1e-7, 1e-6).all()
    assert close(s, gt["act0_scaling"], 1e-9, 1e-6).all()
    assert close(r, gt["act0_rotation"], 1e-7, 1e-6).all()
    feats = np.concatenate([gt["adam_theta0_f_dc"], gt["adam_theta0_f_rest"]], axis=1)
    assert np.array_equal(feats, gt["act0_features"])
    A = T.AdamOracle(*(gt["adam_theta0_" + n] for n in ("xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation")), eps=float(gt["adam_eps"]))
    for it in (1, 2, 3):
        p = A.step(gt["adam_lrs"][it - 1], gt["adam_g%d_xyz" % it], gt["adam_g%d_features" % it], gt["adam_g%d_opacity" % it],
                   gt["adam_g%d_scaling" % it], gt["adam_g%d_rotation" % it])
        for n in ("xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation"):
            ref = gt["adam_theta%d_%s" % (it, n)]
            assert close(p[n], ref, 2e-6, 2e-6).all(), (it, n, np.abs(p[n] - ref).max())


def test_learning_rates_match_reference_setup(gt):
    # scene/gaussian_model.py:153-166 with arguments/__init__.py:75-95 defaults and spatial_lr_scale 2.5
    sl = float(gt["adam_spatial_lr_scale"])
    for it, ref in gt["xyz_lr_at"]:
        mine = T.expon_lr(int(it), 0.00016 * sl, 0.0000016 * sl, lr_delay_mult=0.01, max_steps=30000)
        assert abs(mine - ref) <= 1e-12 + 1e-9 * ref
    lr = gt["adam_lrs"][0]
    assert np.allclose(lr[1:], [0.0025, 0.0025 / 20.0, 0.05, 0.005, 0.001])
    assert float(gt["adam_eps"]) == 1e-15 and tuple(gt["adam_betas"]) == (0.9, 0.999)


def test_densify_stats_match_reference(gt):
    P = int(gt["adam_P"])
    acc, den, mr = np.zeros((P,)), np.zeros((P,)), np.zeros((P,))
    for view in range(2):
        acc, den, mr = T.densify_stats(acc, den, mr, gt["dens_g2d_%d" % view], gt["dens_radii_%d" % view])
    assert close(acc, gt["dens_accum"].reshape(-1), 1e-6, 1e-6).all()
    assert np.array_equal(den, gt["dens_denom"].reshape(-1))
    assert np.array_equal(mr, gt["dens_max_radii"].reshape(-1))


def test_camera_matches_reference_camera():
    """surfel_render.Camera(R, T, FoVx, FoVy) builds the matrices the reference's Camera builds (scene/cameras.py:50-59,
    utils/graphics_utils.py:38-71): checked against the matrices captured from the reference's own Camera in ref_intree.npz."""
    import math
    import torch
    import surfel_render as R
    g = np.load(os.path.join(REPO, "tests", "golden", "ref_intree.npz"))
    w2c = g["viewmatrix"].T.astype(np.float64)
    H, W = int(g["image_height"]), int(g["image_width"])
    cam = R.Camera(colmap_id=0, R=w2c[:3, :3].T, T=w2c[:3, 3], FoVx=2 * math.atan(float(g["tanfovx"])), FoVy=2 * math.atan(float(g["tanfovy"])),
                   image=torch.zeros(3, H, W), data_device="cpu")
    assert np.allclose(cam.world_view_transform.numpy(), g["viewmatrix"], rtol=1e-6, atol=1e-7)
    assert np.allclose(cam.full_proj_transform.numpy(), g["projmatrix"], rtol=1e-5, atol=1e-6)
    assert np.allclose(cam.camera_center.numpy(), g["campos"], rtol=1e-5, atol=1e-6)
    assert cam.image_width == W and cam.image_height == H and cam.znear == 0.01 and cam.zfar == 100.0


def test_rotation_and_splat2world_match_reference(gt):
    """surfel_model.quat_to_rotmat == utils/general_utils.build_rotation; the splat2world matrices the compute_cov3D_python path
    feeds the rasterizer == GaussianModel.get_covariance (scene/gaussian_model.py:27-33), here evaluated with torch on CPU."""
    import torch
    import surfel_model as M
    q = torch.tensor(gt["rot_q"])
    assert np.allclose(M.quat_to_rotmat(q).numpy(), gt["rot_R"], rtol=1e-5, atol=1e-6)
    m = M.GaussianModel.__new__(M.GaussianModel)          # no device store: only the pure-torch covariance builder is exercised
    m.P, m.device = 64, torch.device("cpu")
    m._pv = {"xyz": torch.tensor(gt["cov_xyz"]), "rotation": q}
    m._av = {"scaling": torch.exp(torch.tensor(gt["cov_scaling"])), "rotation": q / q.norm(dim=1, keepdim=True)}      # activated values
    m.grad = None
    assert np.allclose(m.get_covariance(1.7).numpy(), gt["cov_splat2world"], rtol=1e-5, atol=1e-6)
