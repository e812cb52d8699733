"""GPU parity of the training-iteration kernels (include/surfel_train.h) — every call goes through the C ABI.

Checked against (1) tests/golden/ref_train.npz, produced by the reference's own Python (fp32 torch on CPU), and
(2) oracle/train_oracle.py (fp64) on larger seeded inputs.  Stated fp32 tolerances:
  loss means |d| <= 2e-6 (l1) / 2e-5 (ssim); gradients |d| <= 2e-3*mean|ref| + 2e-3*|ref| on >= 99.9 % of elements, cosine >= 0.99999;
  maps 1e-5 + 1e-5*|ref| (surf_normal: 2e-3 on >= 99.5 %: finite differences of fp32 points); Adam parameters 2e-6 + 2e-6*|ref|.
"""
import math
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def gt():
    return np.load(os.path.join(REPO, "tests", "golden", "ref_train.npz"))


def dev():
    import torch
    return torch.device("cuda:0")


def T(a):
    import torch
    return torch.as_tensor(np.ascontiguousarray(a, dtype=np.float32)).to(dev())


def close_frac(a, b, atol, rtol):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return float((np.abs(a - b) <= atol + rtol * np.abs(b)).mean())


def cosine(a, b):
    a = np.asarray(a, np.float64).ravel(); b = np.asarray(b, np.float64).ravel()
    return float(a @ b / (np.linalg.norm(a) * np.linalg.norm(b)))


def grad_ok(mine, ref, frac=0.999, cos=0.99999):
    scale = np.abs(ref).mean()
    f = close_frac(mine, ref, 2e-3 * scale, 2e-3)
    c = cosine(mine, ref)
    assert f >= frac and c >= cos, (f, c)


# ------------------------------------------------------------------------------------------------ losses
@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_l1_ssim_match_reference_golden(gt, tag):
    import torch
    import surfel_losses as L
    img, tgt = gt["loss_%s_img" % tag], gt["loss_%s_gt" % tag]
    x = T(img).requires_grad_(True); y = T(tgt)
    l1 = L.l1_loss(x, y); s = L.ssim(x, y)
    assert abs(float(l1.detach()) - float(gt["loss_%s_l1" % tag])) < 2e-6
    assert abs(float(s.detach()) - float(gt["loss_%s_ssim" % tag])) < 2e-5
    g_l1, = torch.autograd.grad(l1, x); g_s, = torch.autograd.grad(s, x)
    assert close_frac(g_l1.cpu().numpy(), gt["loss_%s_g_l1" % tag], 1e-9, 1e-5) == 1.0      # includes sign(0) = 0 ties
    grad_ok(g_s.cpu().numpy(), gt["loss_%s_g_ssim" % tag])
    # fused form, train.py:72-74
    x2 = T(img).requires_grad_(True)
    loss, means = L.photometric_loss(x2, y, 0.2)
    assert abs(float(loss.detach()) - float(gt["loss_%s_total" % tag])) < 2e-5
    (3.0 * loss).backward()
    grad_ok(x2.grad.cpu().numpy() / 3.0, gt["loss_%s_g_total" % tag])
    assert tuple(x2.grad.shape) == img.shape


def test_l1_ssim_vs_oracle_large_and_reproducible():
    import torch
    import surfel_losses as L
    from oracle import train_oracle as O
    rng = np.random.default_rng(9)
    tgt = rng.uniform(0, 1, size=(3, 203, 331)).astype(np.float32)
    img = np.clip(tgt + 0.1 * rng.normal(size=tgt.shape), 0, 1).astype(np.float32)
    o = O.photometric(img, tgt, 0.2)
    outs = []
    for _ in range(2):
        x = T(img).requires_grad_(True)
        loss, means = L.photometric_loss(x, T(tgt), 0.2)
        loss.backward()
        outs.append((float(loss), means.cpu().numpy().copy(), x.grad.cpu().numpy().copy()))
    assert abs(outs[0][0] - o["loss"]) < 2e-5 and abs(outs[0][1][0] - o["l1"]) < 2e-6 and abs(outs[0][1][1] - o["ssim"]) < 2e-5
    grad_ok(outs[0][2], o["g_loss"])
    assert outs[0][0] == outs[1][0] and np.array_equal(outs[0][2], outs[1][2])        # fixed-order reductions: bit-reproducible
    with pytest.raises(RuntimeError):
        L.l1_loss(torch.zeros(3, 8, 8), torch.zeros(3, 8, 8))                           # CPU tensors: no fallback


# ------------------------------------------------------------------------------------------------ render post-processing
class _View:
    def __init__(self, gt):
        import torch
        self.world_view_transform = T(gt["post_world_view_transform"]); self.full_proj_transform = T(gt["post_full_proj_transform"])
        self.image_width, self.image_height = int(gt["post_W"]), int(gt["post_H"])


@pytest.mark.parametrize("ratio", [0, 1])
def test_render_post_matches_reference_render(gt, ratio):
    import torch
    import surfel_render as R
    view = _View(gt)
    am = T(gt["post_allmap"]).requires_grad_(True)
    maps = R.render_post(am, view, float(ratio))
    ref = gt["post_r%d_maps" % ratio]
    m = maps.detach().cpu().numpy()
    assert close_frac(m[:6], ref[:6], 1e-5, 1e-5) == 1.0
    assert close_frac(m[6:], ref[6:], 2e-3, 1e-3) > 0.995
    (maps * T(gt["post_wmaps"])).sum().backward()
    g, gref = am.grad.cpu().numpy(), gt["post_r%d_g_maps" % ratio]
    ok = np.isfinite(gref)                       # 0/0 pixels: NaN in the reference's autograd, zero here
    assert np.isfinite(g).all() and (g[0][~ok[0]] == 0).all()     # d/d(sum w*depth) is zero where alpha == 0
    scale = np.abs(gref[ok]).mean()
    assert close_frac(g[ok], gref[ok], 2e-3 * scale, 2e-2) > 0.99
    assert cosine(g[ok], gref[ok]) > 0.9999
    # fused regularisers straight from allmap (train.py:80-85)
    am2 = T(gt["post_allmap"]).requires_grad_(True)
    ln, ld = float(gt["post_lambda_normal"]), float(gt["post_lambda_dist"])
    reg, means = R.regularizers(am2, view, float(ratio), ln, ld)
    assert abs(float(means[0]) - float(gt["post_r%d_normal_err_mean" % ratio])) < 1e-4
    assert abs(float(means[1]) - float(gt["post_r%d_dist_mean" % ratio])) < 1e-7
    assert abs(float(reg) - (ln * float(gt["post_r%d_normal_err_mean" % ratio]) + ld * float(gt["post_r%d_dist_mean" % ratio]))) < 1e-5
    reg.backward()
    g, gref = am2.grad.cpu().numpy(), gt["post_r%d_g_reg" % ratio]
    ok = np.isfinite(gref)
    scale = np.abs(gref[ok]).mean()
    assert close_frac(g[ok], gref[ok], 2e-3 * scale, 2e-2) > 0.99
    assert cosine(g[ok], gref[ok]) > 0.9999


def test_render_post_vs_oracle_odd_size():
    """non-multiple-of-16 image, tilted camera; HIP fp32 vs the fp64 restatement, and bitwise equality of the fused
    regulariser gradient with the modular path fed the same upstream gradients."""
    import torch
    import surfel_render as R
    import synthetic
    from oracle import train_oracle as O
    W, H = 77, 53
    sc = synthetic.make_scene(8, W, H, seed=1, view_index=3)
    wvt, fpt = sc["viewmatrix"], sc["projmatrix"]
    rng = np.random.default_rng(2)
    yy, xx = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
    alpha = np.clip(0.7 + 0.3 * np.sin(xx / 9.0) * np.cos(yy / 6.0), 0.05, 0.99).astype(np.float32)
    depth = (4.0 + np.sin(xx / 5.0) + 0.7 * np.cos(yy / 4.0)).astype(np.float32)
    am = np.zeros((7, H, W), np.float32)
    am[0] = depth * alpha; am[1] = alpha
    n = rng.normal(size=(3, H, W)); am[2:5] = (n / np.linalg.norm(n, axis=0)) * alpha
    am[5] = depth * 1.01; am[6] = 0.01 * np.abs(rng.normal(size=(H, W)))
    wm = rng.normal(size=(9, H, W)).astype(np.float32)

    class V: pass
    v = V(); v.world_view_transform = T(wvt); v.full_proj_transform = T(fpt); v.image_width = W; v.image_height = H
    for ratio in (0.0, 1.0, 0.3):
        o = O.render_post_np(am, wvt, fpt, W, H, ratio, wmaps=wm, lambda_normal=0.05, lambda_dist=10.0)
        a = T(am).requires_grad_(True)
        maps = R.render_post(a, v, ratio)
        assert close_frac(maps.detach().cpu().numpy(), o["maps"], 2e-5, 2e-5) > 0.999
        (maps * T(wm)).sum().backward()
        grad_ok(a.grad.cpu().numpy(), o["g_maps"], frac=0.995, cos=0.9999)
        a2 = T(am).requires_grad_(True)
        reg, means = R.regularizers(a2, v, ratio, 0.05, 10.0)
        reg.backward()
        grad_ok(a2.grad.cpu().numpy(), o["g_reg"], frac=0.995, cos=0.9999)
        assert abs(float(means[0]) - o["normal_err_mean"]) < 2e-5 and abs(float(means[1]) - o["dist_mean"]) < 1e-7


def test_camera_consts_match_reference_formulas(gt):
    import surfel_render as R
    from oracle import train_oracle as O
    W, H = int(gt["post_W"]), int(gt["post_H"])
    mine = R.post_consts(gt["post_world_view_transform"], gt["post_full_proj_transform"], W, H)
    ref = O.cam_consts(gt["post_world_view_transform"], gt["post_full_proj_transform"], W, H)
    assert np.allclose(mine, ref, rtol=1e-6, atol=1e-7)


# ------------------------------------------------------------------------------------------------ parameter store
def _model_from_golden(gt):
    import surfel_model as M
    m = M.GaussianModel(3, device=dev())
    m.set_parameters(gt["adam_theta0_xyz"], gt["adam_theta0_f_dc"], gt["adam_theta0_f_rest"], gt["adam_theta0_opacity"],
                     gt["adam_theta0_scaling"], gt["adam_theta0_rotation"])
    return m


def test_activations_and_adam_match_reference_model(gt):
    import torch
    import surfel_trainer as TR
    m = _model_from_golden(gt)
    assert close_frac(m.get_opacity.detach().cpu().numpy(), gt["act0_opacity"], 1e-6, 1e-6) == 1.0
    assert close_frac(m.get_scaling.detach().cpu().numpy(), gt["act0_scaling"], 1e-8, 2e-6) == 1.0
    assert close_frac(m.get_rotation.detach().cpu().numpy(), gt["act0_rotation"], 1e-6, 1e-6) == 1.0
    assert np.array_equal(m.get_features.detach().cpu().numpy(), gt["act0_features"])
    m.spatial_lr_scale = float(gt["adam_spatial_lr_scale"])
    m.training_setup(TR.optimization_params())
    for it in (1, 2, 3):
        lr = m.update_learning_rate(it)
        assert np.allclose(np.array(m.lr, np.float64), gt["adam_lrs"][it - 1], rtol=1e-6)
        gv = m._gv
        gv["xyz"].copy_(T(gt["adam_g%d_xyz" % it])); gv["sh"].copy_(T(gt["adam_g%d_features" % it]).reshape(m.P, 48))
        gv["opacity"].copy_(T(gt["adam_g%d_opacity" % it])); gv["scaling"].copy_(T(gt["adam_g%d_scaling" % it]))
        gv["rotation"].copy_(T(gt["adam_g%d_rotation" % it]))
        m.optimizer_step()
        for name, mine in (("xyz", m._xyz), ("f_dc", m._features_dc), ("f_rest", m._features_rest), ("opacity", m._opacity),
                           ("scaling", m._scaling), ("rotation", m._rotation)):
            ref = gt["adam_theta%d_%s" % (it, name)]
            assert close_frac(mine.cpu().numpy(), ref, 2e-6, 2e-6) == 1.0, (it, name, np.abs(mine.cpu().numpy() - ref).max())
        # activations are refreshed by the optimiser kernel itself
        assert close_frac(m.get_opacity.detach().cpu().numpy(), 1 / (1 + np.exp(-gt["adam_theta%d_opacity" % it].astype(np.float64))), 1e-6, 1e-6) == 1.0
        q = gt["adam_theta%d_rotation" % it].astype(np.float64)
        assert close_frac(m.get_rotation.detach().cpu().numpy(), q / np.linalg.norm(q, axis=1, keepdims=True), 2e-6, 2e-6) == 1.0
        assert close_frac(m.get_scaling.detach().cpu().numpy(), np.exp(gt["adam_theta%d_scaling" % it].astype(np.float64)), 1e-8, 4e-6) == 1.0


def test_adam_vs_oracle_many_surfels():
    import surfel_model as M
    import surfel_trainer as TR
    from oracle import train_oracle as O
    rng = np.random.default_rng(4)
    P = 5003
    th = dict(xyz=rng.normal(size=(P, 3)), f_dc=rng.normal(size=(P, 1, 3)), f_rest=0.1 * rng.normal(size=(P, 15, 3)),
              opacity=rng.normal(0, 2, size=(P, 1)), scaling=rng.normal(-3, 1, size=(P, 2)), rotation=rng.normal(size=(P, 4)))
    th = {k: v.astype(np.float32) for k, v in th.items()}
    m = M.GaussianModel(3, device=dev())
    m.set_parameters(th["xyz"], th["f_dc"], th["f_rest"], th["opacity"], th["scaling"], th["rotation"])
    m.spatial_lr_scale = 1.0
    m.training_setup(TR.optimization_params())
    A = O.AdamOracle(th["xyz"], th["f_dc"], th["f_rest"], th["opacity"], th["scaling"], th["rotation"])
    for it in range(1, 6):
        m.update_learning_rate(it)
        g = dict(xyz=rng.normal(size=(P, 3)) * 1e-3, sh=rng.normal(size=(P, 16, 3)) * 1e-3, opacity=rng.normal(size=(P, 1)) * 1e-2,
                 scaling=rng.normal(size=(P, 2)) * 1e-2, rotation=rng.normal(size=(P, 4)) * 1e-3)
        g = {k: v.astype(np.float32) for k, v in g.items()}
        for k in g:
            m._gv[k].copy_(T(g[k]).reshape(m._gv[k].shape))
        m._gv["xyz"].mul_(2.0)                       # grad_scale path: (2 g) * 0.5
        m._gv["sh"].mul_(2.0); m._gv["opacity"].mul_(2.0); m._gv["scaling"].mul_(2.0); m._gv["rotation"].mul_(2.0)
        m.optimizer_step(grad_scale=0.5)
        p = A.step(m.lr, g["xyz"], g["sh"], g["opacity"], g["scaling"], g["rotation"])
        for name, mine in (("xyz", m._xyz), ("f_dc", m._features_dc), ("f_rest", m._features_rest), ("opacity", m._opacity),
                           ("scaling", m._scaling), ("rotation", m._rotation)):
            assert close_frac(mine.cpu().numpy(), p[name], 3e-6, 3e-6) == 1.0, (it, name)


def test_densify_stats_match_reference(gt):
    import torch
    m = _model_from_golden(gt)
    import surfel_trainer as TR
    m.training_setup(TR.optimization_params())
    for view in range(2):
        m.add_densification_stats(T(gt["dens_g2d_%d" % view]), radii=torch.as_tensor(gt["dens_radii_%d" % view]).to(dev()))
    assert close_frac(m.xyz_gradient_accum.cpu().numpy(), gt["dens_accum"], 1e-6, 1e-6) == 1.0
    assert np.array_equal(m.denom.cpu().numpy(), gt["dens_denom"])
    assert np.array_equal(m.max_radii2D.cpu().numpy(), gt["dens_max_radii"])


def test_densify_prune_reset_bookkeeping():
    """Clone / split / prune on the flat store: counts, carried Adam moments, statistics reset (gaussian_model.py:329-403)."""
    import torch
    import surfel_model as M
    import surfel_trainer as TR
    torch.manual_seed(0)
    P = 1000
    m = TR.synthetic_object(P, dev(), seed=2, px_scale=0.05)
    m.spatial_lr_scale = 1.0
    opt = TR.optimization_params()
    m.training_setup(opt)
    mm = M._views(m.m, P)
    mm["xyz"].copy_(torch.arange(P, device=dev(), dtype=torch.float32)[:, None].expand(P, 3))     # tag rows to follow them
    extent = 5.0
    sc = m.get_scaling.max(dim=1).values
    big = sc > opt.percent_dense * extent
    m.xyz_gradient_accum[:] = 0.0; m.denom[:] = 1.0
    hot = torch.zeros(P, dtype=torch.bool, device=dev()); hot[::3] = True
    m.xyz_gradient_accum[hot] = 1.0
    n_clone = int((hot & ~big).sum()); n_split = int((hot & big).sum())
    low = (m.get_opacity.squeeze() < 0.5)
    xyz_before = m._xyz.clone()
    m.densify_and_prune(0.5, 0.5, extent, None)
    # clones are appended; split originals are removed and replaced by 2 samples each; then low-opacity surfels are pruned
    n_low_after = int(low.sum()) + int((low & hot & ~big).sum()) + int((low & hot & big).sum())   # clones / samples inherit opacity
    assert m.P == P + n_clone + n_split - n_low_after
    assert m.xyz_gradient_accum.shape == (m.P, 1) and float(m.xyz_gradient_accum.abs().sum()) == 0 and float(m.denom.sum()) == 0
    assert m.max_radii2D.shape == (m.P,)
    assert (m.get_opacity >= 0.5).all()
    tags = M._views(m.m, m.P)["xyz"][:, 0]
    kept = tags[tags > 0].long()                       # surviving original rows keep their Adam moments, new rows start at zero
    assert torch.equal(m._xyz[tags > 0], xyz_before[kept])
    assert m.grad.numel() == m.P * 58 and m.theta.numel() == m.P * 58
    # activations were refreshed for the new store
    assert close_frac(m.get_scaling.detach().cpu().numpy(), np.exp(m._scaling.cpu().numpy().astype(np.float64)), 1e-8, 4e-6) == 1.0
    m.reset_opacity()
    assert float(m.get_opacity.max()) <= 0.0100001
    assert float(M._views(m.m, m.P)["opacity"].abs().sum()) == 0


def test_checkpoint_and_ply_roundtrip(tmp_path):
    import torch
    import surfel_model as M
    import surfel_trainer as TR
    m = TR.synthetic_object(300, dev(), seed=5)
    m.spatial_lr_scale = 2.0
    opt = TR.optimization_params()
    m.training_setup(opt)
    m.grad.normal_(0, 1e-3)
    m.update_learning_rate(1); m.optimizer_step()
    ck = m.capture()
    torch.save((ck, 1), tmp_path / "chkpnt1.pth")
    (args, it) = torch.load(tmp_path / "chkpnt1.pth", weights_only=False)
    m2 = M.GaussianModel(3, device=dev())
    m2.restore(args, opt)
    assert it == 1 and m2.step_count == 1
    assert torch.equal(m2.theta, m.theta) and torch.equal(m2.m, m.m) and torch.equal(m2.v, m.v)
    assert set(ck[10]["state"][0].keys()) == {"step", "exp_avg", "exp_avg_sq"} and [g["name"] for g in ck[10]["param_groups"]] == list(M.GROUPS)
    m.save_ply(str(tmp_path / "point_cloud.ply"))
    m3 = M.GaussianModel(3, device=dev())
    m3.load_ply(str(tmp_path / "point_cloud.ply"))
    assert torch.equal(m3.theta, m.theta) and m3.active_sh_degree == 3


# ------------------------------------------------------------------------------------------------ render() + training
def test_render_dict_matches_reference_contract():
    import torch
    import surfel_render as R
    import surfel_trainer as TR
    gtm = TR.synthetic_object(3000, dev(), seed=1, px_scale=0.06)
    cams = TR.orbit_cameras(2, 96, 80, device=dev())
    bg = torch.zeros(3, device=dev())
    out = R.render(cams[0], gtm, TR.pipeline_params(depth_ratio=1.0), bg)
    assert set(out) >= {"render", "viewspace_points", "visibility_filter", "radii", "rend_alpha", "rend_normal", "rend_dist", "surf_depth",
                        "surf_normal"}
    assert out["render"].shape == (3, 80, 96) and out["rend_alpha"].shape == (1, 80, 96) and out["rend_normal"].shape == (3, 80, 96)
    assert out["surf_normal"].shape == (3, 80, 96) and out["visibility_filter"].dtype == torch.bool
    assert int(out["visibility_filter"].sum()) > 1000 and float(out["rend_alpha"].max()) > 0.5
    am = out["allmap"]
    assert torch.equal(out["rend_alpha"][0], am[1]) and torch.equal(out["rend_dist"][0], am[6])
    # surf_normal is a unit vector times alpha wherever it is defined
    nrm = out["surf_normal"].norm(dim=0)
    inner = nrm[1:-1, 1:-1]; a = am[1][1:-1, 1:-1]
    sel = a > 0.05
    assert torch.allclose(inner[sel], a[sel], rtol=1e-4, atol=1e-5)
    assert float(nrm[0].abs().max()) == 0 and float(nrm[:, 0].abs().max()) == 0


def test_training_converges_and_densifies():
    """The loop of train.py on the fused kernels: PSNR on the training views must rise markedly from a perturbed start, with
    regularisers on and densification / opacity reset exercised."""
    import torch
    import surfel_trainer as TR
    torch.manual_seed(0)
    d = dev()
    bg = torch.zeros(3, device=d)
    gtm = TR.synthetic_object(2500, d, seed=3, px_scale=0.07)
    cams = TR.capture_views(gtm, TR.orbit_cameras(10, 112, 96, device=d), bg)
    # start: ground-truth positions jittered, everything else reset the way create_from_pcd initialises
    g = torch.Generator().manual_seed(1)
    pts = (gtm._xyz.cpu() + 0.03 * torch.randn((gtm.P, 3), generator=g)).numpy()
    pcd = type("PCD", (), {})(); pcd.points = pts; pcd.colors = np.full((gtm.P, 3), 0.5, np.float32)
    import surfel_model as M
    model = M.GaussianModel(3, device=d)
    model.create_from_pcd(pcd, spatial_lr_scale=TR.cameras_extent(cams))
    opt = TR.optimization_params(iterations=400, densify_from_iter=100, densification_interval=100, densify_until_iter=350,
                                 opacity_reset_interval=300, dist_from_iter=50, normal_from_iter=100, lambda_dist=10.0, lambda_normal=0.05,
                                 position_lr_max_steps=400)
    tr = TR.Trainer(model, cams, opt, TR.pipeline_params(depth_ratio=1.0))
    p0, _ = tr.evaluate()
    P0 = model.P
    for _ in range(opt.iterations):
        tr.step()
    p1, l1 = tr.evaluate()
    print('PSNR %.2f -> %.2f dB, points %d -> %d' % (p0, p1, P0, model.P))
    assert math.isfinite(p1) and p1 > p0 + 3.0, (p0, p1)
    assert model.P != P0                      # densification / pruning happened
    assert torch.isfinite(model.theta).all() and model.step_count < opt.iterations


@pytest.mark.parametrize("ratio", [0, 1])
def test_single_node_train_loss_matches_reference_sum(gt, ratio):
    """train.py:72-88 as one autograd node: value and both gradients equal the reference's separately computed pieces."""
    import torch
    import surfel_losses as L
    import surfel_render as R
    view = _View(gt)
    W, H = int(gt["post_W"]), int(gt["post_H"])
    rng = np.random.default_rng(3)
    tgt = rng.uniform(0, 1, size=(3, H, W)).astype(np.float32)
    img = np.clip(tgt + 0.1 * rng.normal(size=tgt.shape), 0, 1).astype(np.float32)
    from oracle import train_oracle as O
    o = O.photometric(img, tgt, 0.2)
    ln, ld = float(gt["post_lambda_normal"]), float(gt["post_lambda_dist"])
    x = T(img).requires_grad_(True); am = T(gt["post_allmap"]).requires_grad_(True)
    cam = T(R.post_consts(gt["post_world_view_transform"], gt["post_full_proj_transform"], W, H))
    total, sc = L.train_loss(x, am, T(tgt), cam, float(ratio), 0.2, ln, ld)
    sc = sc.cpu().numpy()
    ne, di = float(gt["post_r%d_normal_err_mean" % ratio]), float(gt["post_r%d_dist_mean" % ratio])
    assert abs(sc[0] - o["l1"]) < 2e-6 and abs(sc[1] - o["ssim"]) < 2e-5 and abs(sc[2] - ne) < 1e-4 and abs(sc[3] - di) < 1e-7
    assert abs(sc[4] - o["loss"]) < 2e-5 and abs(float(total.detach()) - (o["loss"] + ln * ne + ld * di)) < 3e-5 and sc[5] == float(total.detach())
    (2.0 * total).backward()
    grad_ok(x.grad.cpu().numpy() / 2.0, o["g_loss"])
    g, gref = am.grad.cpu().numpy() / 2.0, gt["post_r%d_g_reg" % ratio]
    ok = np.isfinite(gref)
    assert close_frac(g[ok], gref[ok], 2e-3 * np.abs(gref[ok]).mean(), 2e-2) > 0.99 and cosine(g[ok], gref[ok]) > 0.9999
    # no regularisers: allmap is not touched
    x2 = T(img).requires_grad_(True)
    t2, sc2 = L.train_loss(x2, None, T(tgt), None, 0.0, 0.2, 0.0, 0.0)
    assert abs(float(t2.detach()) - o["loss"]) < 2e-5 and float(sc2[2]) == 0.0
    t2.backward()
    grad_ok(x2.grad.cpu().numpy(), o["g_loss"])


@pytest.mark.parametrize("hw", [(64, 80), (101, 77), (800, 800)])
def test_fused_loss_launches_keep_the_bits(hw):
    """The training loss's two halves in one launch per direction (csrc/train_fused.hip: SSIM workgroups and post-processing
    workgroups of one grid, the same kernel bodies over one LDS workspace; the loss scalars by a finalize launch or by an extra
    workgroup of the backward launch) against the four separate launches: loss scalars and both gradients BIT-IDENTICAL, also on frame sizes
    that are no multiple of either tile."""
    import torch
    import surfel_losses as L
    import surfel_render as R
    import surfel_trainer as TR
    H, W = hw
    d = dev()
    cams = TR.orbit_cameras(1, W, H, device=d)
    cam = cams[0]
    g = torch.Generator().manual_seed(H * 1000 + W)
    tgt = torch.rand((3, H, W), generator=g).to(d)
    img = (tgt + 0.1 * torch.randn((3, H, W), generator=g).to(d)).clamp(0, 1)
    allmap = torch.randn((7, H, W), generator=g).to(d)
    allmap[0] = allmap[0].abs() * 3 + 0.5; allmap[1] = torch.rand((H, W), generator=g).to(d) * 0.9 + 0.05; allmap[5] = allmap[5].abs() * 3 + 0.5
    consts = cam.post_consts()
    res = []
    try:
        # separate launches | fused halves + finalize launch | fused halves, loss scalars deferred to an extra workgroup of the backward
        for fused, defer in ((False, False), (True, False), (True, True), (False, True)):
            L.FUSED_LOSS = fused
            x = img.clone().requires_grad_(True); am = allmap.clone().requires_grad_(True)
            total, sc = L.train_loss(x, am, tgt, consts, 0.7, 0.2, 0.05, 100.0, defer_scalars=defer)
            total.backward(torch.full((), 1.5, device=d))
            res.append((total.detach().clone(), sc.clone(), x.grad.clone(), am.grad.clone()))
    finally:
        L.FUSED_LOSS = True
    for other in res[1:]:
        for a, b in zip(res[0], other):
            assert torch.isfinite(a).all() and torch.equal(a, b)


def test_trainer_with_lazily_counted_forwards_is_identical():
    """Trainer.lazy_count: the iteration without the host wait for the instance count (forward returns its capacity, the count is
    collected after the backward) trains to the same bits as the waiting form — also when a frame is reported as overflowed and the
    iteration's forward, loss and backward are redone (forced here: the collection is made to fail once).  Trainer.manual_chain: the
    iteration's forward / backward chain driven by hand (surfel_native.ManualCtx) against torch.autograd driving it: same bits."""
    import torch
    import diff_surfel_rasterization as dsr
    import surfel_native as n
    import surfel_trainer as TR
    d = dev()
    bg = torch.zeros(3, device=d)
    gt_model = TR.synthetic_object(3000, d, seed=1, px_scale=0.06)
    cams = TR.capture_views(gt_model, TR.orbit_cameras(4, 144, 96, device=d), bg)
    out = []
    real = dsr.finish_count
    for mode in ("wait", "lazy", "lazy+redo", "wait+autograd", "lazy+autograd"):
        m = TR.synthetic_object(3000, d, seed=2, px_scale=0.05)
        m.spatial_lr_scale = 1.0
        tr = TR.Trainer(m, cams, TR.optimization_params(dist_from_iter=2, normal_from_iter=0, lambda_dist=10.0, densify_from_iter=10 ** 9),
                        TR.pipeline_params(depth_ratio=1.0))
        tr.lazy_count = not mode.startswith("wait")
        tr.manual_chain = "autograd" not in mode      # the chain driven by hand (default) | through torch.autograd
        calls = [0]

        def fake():
            calls[0] += 1
            r = real()
            if calls[0] == 3:
                raise n.CapacityOverflow("forced")
            return r
        if mode == "lazy+redo":
            dsr.finish_count = fake
        try:
            scal = []
            for _ in range(5):
                tr.step()
                scal.append(tr.last["scalars"].clone())
        finally:
            dsr.finish_count = real
        torch.cuda.synchronize()
        assert tr.lazy_overflows == (1 if mode == "lazy+redo" else 0)
        if mode.startswith("lazy"):
            assert n.load().surfel_debug_last_binning() in (0, 4)
        out.append((m.theta.clone(), m.m.clone(), m.v.clone(), m.xyz_gradient_accum.clone(), m.denom.clone(), torch.stack(scal)))
    for other in out[1:]:
        for a, b in zip(out[0], other):
            assert torch.equal(a, b)


def test_fused_update_equals_the_three_launches():
    """surfel_train_update (densification statistics + SH Adam + geometry Adam in ONE launch) against the three kernels it replaces:
    parameters, moments, activations and statistics equal to the bit over iterations that include a densification (where the trainer
    falls back to the separate calls) — and the fused entry point itself against the separate ones on one state."""
    import torch
    import surfel_trainer as TR
    d = dev()
    bg = torch.zeros(3, device=d)
    gt_model = TR.synthetic_object(2500, d, seed=4, px_scale=0.06)
    cams = TR.capture_views(gt_model, TR.orbit_cameras(4, 128, 96, device=d), bg)
    out = []
    for fused in (True, False):
        torch.manual_seed(77)      # (the split samples of densify_and_split come from the global generator)
        m = TR.synthetic_object(2500, d, seed=5, px_scale=0.05)
        m.spatial_lr_scale = 1.0
        tr = TR.Trainer(m, cams, TR.optimization_params(dist_from_iter=2, normal_from_iter=0, lambda_dist=10.0, densify_from_iter=3, densification_interval=4,
                                                        densify_until_iter=12, opacity_reset_interval=10), TR.pipeline_params(depth_ratio=1.0))
        tr.fused_update = fused
        for _ in range(15):      # densifies at 4 and 8, resets the opacities at 10, statistics off from 12
            tr.step()
        torch.cuda.synchronize()
        out.append((m.P, m.theta.clone(), m.m.clone(), m.v.clone(), m.act.clone(), m.xyz_gradient_accum.clone(), m.denom.clone(), m.max_radii2D.clone()))
    assert out[0][0] == out[1][0] and out[0][0] > 1000
    for a, b in zip(out[0][1:], out[1][1:]):
        assert torch.equal(a, b)


def test_trainer_redoes_a_lazily_counted_frame_that_really_overflowed():
    """A frame that REALLY overflows its binning capacity inside Trainer.step (VERDICT r3 weak #2 / ADVICE r3 high): the capacity of
    this frame size was learnt on a 3 000-surfel model, then a 4x larger model trains at the same size.  Its first lazily counted
    forward returns the old capacity, loss and backward are enqueued on the truncated frame (the backward kernels return at once:
    the gradient records sized from the capacity are far too few), finish_count() reports the overflow and the iteration is redone
    with exact sizes — same bits as a trainer that waits for every count."""
    import torch
    import surfel_native as n
    import surfel_trainer as TR
    d = dev()
    bg = torch.zeros(3, device=d)
    W, H = 176, 112      # (a frame size no other test uses: the capacity history of this thread is this test's own)
    gt_model = TR.synthetic_object(3000, d, seed=1, px_scale=0.06)
    cams = TR.capture_views(gt_model, TR.orbit_cameras(4, W, H, device=d), bg)

    def trainer(points, lazy):
        m = TR.synthetic_object(points, d, seed=2, px_scale=0.06)
        m.spatial_lr_scale = 1.0
        tr = TR.Trainer(m, cams, TR.optimization_params(dist_from_iter=0, normal_from_iter=0, lambda_dist=10.0, densify_from_iter=10 ** 9),
                        TR.pipeline_params(depth_ratio=1.0))
        tr.lazy_count = lazy
        return m, tr

    _, small = trainer(3000, True)
    for _ in range(3):
        small.step()
    assert small.lazy_overflows == 0 and n.load().surfel_debug_last_binning() == 4      # capacity path, lazily counted
    out = []
    for lazy in (True, False):       # the lazy run first: the waiting run would teach the capacity table the larger count
        m, tr = trainer(12000, lazy)
        scal = []
        for _ in range(3):
            tr.step()
            scal.append(tr.last["scalars"].clone())
        torch.cuda.synchronize()
        assert tr.lazy_overflows == (1 if lazy else 0), tr.lazy_overflows
        assert torch.isfinite(torch.stack(scal)).all() and torch.isfinite(m.theta).all()
        out.append((m.theta.clone(), m.m.clone(), m.v.clone(), m.xyz_gradient_accum.clone(), m.denom.clone(), torch.stack(scal)))
    for a, b in zip(out[0], out[1]):
        assert torch.equal(a, b)


def test_render_python_covariance_and_override_color_paths():
    """render()'s compute_cov3D_python branch (gaussian_renderer/__init__.py:59-75 with scene/gaussian_model.py:27-33) and
    override_color produce the same image as the native scale/rotation + SH path."""
    import torch
    import surfel_render as R
    import surfel_trainer as TR
    m = TR.synthetic_object(2000, dev(), seed=4, px_scale=0.06)
    cam = TR.orbit_cameras(1, 88, 72, device=dev())[0]
    bg = torch.tensor([0.1, 0.2, 0.3], device=dev())
    torch.set_grad_enabled(False)
    a = R.render(cam, m, TR.pipeline_params(), bg)
    b = R.render(cam, m, TR.pipeline_params(compute_cov3D_python=True), bg)
    assert close_frac(b["render"].cpu().numpy(), a["render"].cpu().numpy(), 2e-4, 1e-3) > 0.995
    assert close_frac(b["rend_alpha"].cpu().numpy(), a["rend_alpha"].cpu().numpy(), 2e-4, 1e-3) > 0.995
    assert float((a["radii"] > 0).float().mean()) > 0.3
    col = torch.rand((m.P, 3), device=dev())
    c = R.render(cam, m, TR.pipeline_params(), bg, override_color=col)
    assert c["render"].shape == a["render"].shape and not torch.allclose(c["render"], a["render"])
    assert torch.equal(c["radii"], a["radii"]) and torch.allclose(c["rend_alpha"], a["rend_alpha"], atol=1e-6)
    # scaling_modifier shrinks footprints (viewer path, gaussian_renderer/__init__.py:19,42)
    d = R.render(cam, m, TR.pipeline_params(), bg, scaling_modifier=0.5)
    torch.set_grad_enabled(True)
    assert float(d["rend_alpha"].sum()) < float(a["rend_alpha"].sum())


def test_sh_gradient_rebuilt_from_gathered_colour_gradients():
    """View-parallel exchange: sum over views of the rasterizer's dL/dSH == surfel_sh_grad_gather(all views' clamp-masked
    dL/dcolour, camera centres) — the identity that lets N ranks all-gather 12 B/surfel instead of all-reducing 192 B/surfel."""
    import torch
    import surfel_model as M
    import surfel_trainer as TR
    d = dev()
    m = TR.synthetic_object(3000, d, seed=6, px_scale=0.06)
    m._pv["sh"].view(m.P, 16, 3)[:, 0] -= 1.5          # push many colours below zero so the forward's clamp is active
    m.spatial_lr_scale = 1.0
    m.training_setup(TR.optimization_params())
    cams = TR.orbit_cameras(3, 96, 80, device=d)
    bg = torch.zeros(3, device=d)
    for deg in (3, 1):
        m.active_sh_degree = deg
        total_sh = torch.zeros((m.P, 16, 3), device=d); gcols = []; total_xyz = torch.zeros((m.P, 3), device=d)
        for cam in cams:
            m.bind()
            img, radii, allmap, m2 = __import__("surfel_render").rasterize(cam, m, TR.pipeline_params(), bg)
            g = torch.Generator().manual_seed(int(cam.uid) + 10)
            (img * torch.randn(img.shape, generator=g).to(d)).sum().backward()
            total_sh += m._gv["sh"].view(m.P, 16, 3); total_xyz += m._gv["xyz"]
            gcols.append(m.gcol.clone())
        assert float((torch.stack(gcols) == 0).float().mean()) > 0.05          # clamp / culling zeros present
        campos = torch.stack([c.camera_center for c in cams])
        m.sh_grad_from_colours(campos, torch.stack(gcols))
        rebuilt = m._gv["sh"].view(m.P, 16, 3)
        scale = float(total_sh.abs().mean())
        assert torch.allclose(rebuilt, total_sh, rtol=1e-5, atol=1e-6 * scale + 1e-12), float((rebuilt - total_sh).abs().max())
        if deg < 3:
            assert float(rebuilt[:, (deg + 1) ** 2:].abs().max()) == 0.0
    # one view: the rebuilt block equals the rasterizer's own SH gradient
    m.bind()
    img, radii, allmap, m2 = __import__("surfel_render").rasterize(cams[0], m, TR.pipeline_params(), bg)
    img.sum().backward()
    own = m._gv["sh"].clone()
    m.sh_grad_from_colours(cams[0].camera_center[None], m.gcol[None])
    assert torch.allclose(m._gv["sh"], own, rtol=1e-6, atol=1e-9)


def test_fused_sh_adam_equals_explicit_sh_gradients():
    """optimizer_step(colour_grads=...) (SH gradients rebuilt in registers, never written to HBM) takes exactly the step that
    the explicit path takes (rasterizer writes dL/dSH, Adam reads it) — one view and a two-view sum."""
    import torch
    import surfel_render as R
    import surfel_trainer as TR
    d = dev()
    cams = TR.orbit_cameras(2, 96, 80, device=d)
    bg = torch.zeros(3, device=d)

    def make():
        m = TR.synthetic_object(2500, d, seed=8, px_scale=0.06)
        m._pv["sh"].view(m.P, 16, 3)[:, 0] -= 1.0
        m.spatial_lr_scale = 1.0
        m.training_setup(TR.optimization_params())
        return m

    def backward(m, cam, sh_grad):
        m.bind(sh_grad=sh_grad)
        img, radii, allmap, m2 = R.rasterize(cam, m, TR.pipeline_params(), bg)
        g = torch.Generator().manual_seed(int(cam.uid) + 3)
        (img * torch.randn(img.shape, generator=g).to(d)).sum().backward()

    a, b = make(), make()
    for it in (1, 2):
        for m in (a, b):
            m.update_learning_rate(it)
        backward(a, cams[0], True); a.optimizer_step()
        b.grad.fill_(float("nan")); backward(b, cams[0], False)
        # the SH block was not written — except its first 3P floats, where the fused mode keeps the colour gradients (one
        # contiguous 52 B/surfel exchange buffer behind the geometry prefix)
        assert torch.isnan(b._gv["sh"].reshape(-1)[3 * b.P:]).all() and not torch.isnan(b._gv["xyz"]).any()
        assert b.gcol.data_ptr() == b._gv["sh"].data_ptr() and not torch.isnan(b.gcol).any()
        b.optimizer_step(colour_grads=(cams[0].camera_center[None], b.gcol[None]))
        assert torch.allclose(a.theta, b.theta, rtol=1e-5, atol=1e-7), float((a.theta - b.theta).abs().max())
        assert torch.allclose(a.m, b.m, rtol=1e-5, atol=1e-9) and torch.allclose(a.act, b.act, rtol=1e-5, atol=1e-7)
    # two views summed (what two ranks hold after the exchange), averaged
    a, b = make(), make()
    a.update_learning_rate(1); b.update_learning_rate(1)
    tot = torch.zeros_like(a.grad); gc = []
    for cam in cams:
        backward(a, cam, True); tot += a.grad; gc.append(a.gcol.clone())
    a.grad.copy_(tot); a.optimizer_step(grad_scale=0.5)
    b.grad.copy_(tot); b._gv["sh"].fill_(float("nan"))
    b.optimizer_step(grad_scale=0.5, colour_grads=(torch.stack([c.camera_center for c in cams]), torch.stack(gc)))
    assert torch.allclose(a.theta, b.theta, rtol=1e-5, atol=1e-7), float((a.theta - b.theta).abs().max())


def test_resume_from_reference_checkpoint(gt):
    """A chkpnt*.pth written by the REFERENCE (torch.save((gaussians.capture(), iteration)), train.py:142-144; generated by
    tests/golden/make_golden_train.py after two of its iterations) restores into the flat store — parameters, Adam moments, step
    count, learning rates — and the third iteration then lands exactly where the reference's third iteration landed."""
    import torch
    import surfel_model as M
    import surfel_trainer as TR
    (args, it) = torch.load(os.path.join(REPO, "tests", "golden", "ref_chkpnt2.pth"), weights_only=False, map_location="cpu")
    assert it == 2
    m = M.GaussianModel(3, device=dev())
    m.restore(args, TR.optimization_params())
    assert m.step_count == 2 and m.P == int(gt["adam_P"]) and abs(m.spatial_lr_scale - float(gt["adam_spatial_lr_scale"])) < 1e-12
    for name, mine in (("xyz", m._xyz), ("f_dc", m._features_dc), ("f_rest", m._features_rest), ("opacity", m._opacity),
                       ("scaling", m._scaling), ("rotation", m._rotation)):
        assert np.array_equal(mine.cpu().numpy(), gt["adam_theta2_%s" % name]), name
    assert float(m.m.abs().sum()) > 0 and float(m.v.abs().sum()) > 0
    m.update_learning_rate(3)
    assert np.allclose(np.array(m.lr, np.float64), gt["adam_lrs"][2], rtol=1e-6)
    gv = m._gv
    gv["xyz"].copy_(T(gt["adam_g3_xyz"])); gv["sh"].copy_(T(gt["adam_g3_features"]).reshape(m.P, 48))
    gv["opacity"].copy_(T(gt["adam_g3_opacity"])); gv["scaling"].copy_(T(gt["adam_g3_scaling"])); gv["rotation"].copy_(T(gt["adam_g3_rotation"]))
    m.optimizer_step()
    for name, mine in (("xyz", m._xyz), ("f_dc", m._features_dc), ("f_rest", m._features_rest), ("opacity", m._opacity),
                       ("scaling", m._scaling), ("rotation", m._rotation)):
        ref = gt["adam_theta3_%s" % name]
        assert close_frac(mine.cpu().numpy(), ref, 2e-6, 2e-6) == 1.0, (name, np.abs(mine.cpu().numpy() - ref).max())


def test_densify_and_prune_matches_reference_model(gt):
    """scene/gaussian_model.py:348-403 executed by the REFERENCE's GaussianModel on CPU (golden) vs the flat-store implementation:
    same surviving / cloned / split rows in the same order, same copied attributes, same shrunken scales, Adam moments carried
    for surviving rows and zero for new ones, statistics reset.  Only the split samples' positions are random (different RNG
    streams on CPU and GPU): they are checked to lie in their parent's disc plane within a few sigma."""
    import torch
    import surfel_model as M
    import surfel_trainer as TR
    b = {k: gt["dens2_before_" + k] for k in ("xyz", "features_dc", "features_rest", "scaling", "rotation", "opacity")}
    m = M.GaussianModel(3, device=dev())
    m.set_parameters(b["xyz"], b["features_dc"], b["features_rest"], b["opacity"], b["scaling"], b["rotation"])
    m.spatial_lr_scale = 1.0
    opt = TR.optimization_params()
    assert opt.percent_dense == float(gt["dens2_percent_dense"])
    m.training_setup(opt)
    M._views(m.m, m.P)["xyz"].copy_(T(gt["dens2_before_m_xyz"]))
    m.xyz_gradient_accum = T(gt["dens2_in_accum"]); m.denom = T(gt["dens2_in_denom"]); m.max_radii2D = T(gt["dens2_in_max_radii"])
    torch.manual_seed(0)
    m.densify_and_prune(float(gt["dens2_max_grad"]), float(gt["dens2_min_opacity"]), float(gt["dens2_extent"]), int(gt["dens2_max_screen"]))
    assert m.P == int(gt["dens2_P_after"])
    a = {k: gt["dens2_after_" + k] for k in ("xyz", "features_dc", "features_rest", "scaling", "rotation", "opacity")}
    assert np.array_equal(m._features_dc.cpu().numpy(), a["features_dc"]) and np.array_equal(m._features_rest.cpu().numpy(), a["features_rest"])
    assert np.array_equal(m._rotation.cpu().numpy(), a["rotation"]) and np.array_equal(m._opacity.cpu().numpy(), a["opacity"])
    assert close_frac(m._scaling.cpu().numpy(), a["scaling"], 2e-6, 2e-6) == 1.0
    xyz = m._xyz.cpu().numpy(); mom = M._views(m.m, m.P)["xyz"].cpu().numpy()
    ref_m = gt["dens2_after_m_xyz"]
    fixed = np.all(xyz == a["xyz"], axis=1)                                     # rows whose position is not a random sample
    n_rand = int((~fixed).sum())
    assert 0 < n_rand < m.P and n_rand % 2 == 0
    assert np.array_equal(mom[fixed], ref_m[fixed]) and float(np.abs(mom[~fixed]).sum()) == 0 and float(np.abs(ref_m[~fixed]).sum()) == 0
    # the random rows are exactly the reference's split samples: their scales are the parents' divided by 1.6, and both
    # implementations put them within a few sigma of the same parent position
    sig = np.exp(a["scaling"][~fixed]).max(axis=1) * 1.6
    assert (np.linalg.norm(xyz[~fixed] - a["xyz"][~fixed], axis=1) < 12.0 * sig + 1e-6).all()
    assert float(m.xyz_gradient_accum.abs().sum()) == 0 and float(m.denom.sum()) == 0 and float(m.max_radii2D.sum()) == 0


def test_adam_in_two_parts_equals_one_call():
    """surfel_adam_step(parts=1) then (parts=2) — what the view-parallel trainer issues around its two collectives — is the same
    step as one call (parts=3), with explicit SH gradients and with SH gradients rebuilt from colour gradients."""
    import torch
    import surfel_trainer as TR
    d = dev()
    cam = TR.orbit_cameras(1, 64, 48, device=d)[0]

    def make():
        m = TR.synthetic_object(1500, d, seed=9, px_scale=0.06)
        m.spatial_lr_scale = 1.0
        m.training_setup(TR.optimization_params())
        g = torch.Generator().manual_seed(2)
        m.grad.copy_(torch.randn(m.grad.shape, generator=g).to(d) * 1e-3)
        m.gcol.copy_(torch.randn(m.gcol.shape, generator=g).to(d) * 1e-3)
        return m
    for fused in (False, True):
        a, b = make(), make()
        cg = (cam.camera_center[None], a.gcol[None]) if fused else None
        for it in (1, 2):
            a.update_learning_rate(it); b.update_learning_rate(it)
            a.optimizer_step(grad_scale=0.5, colour_grads=cg)
            b.optimizer_step(grad_scale=0.5, colour_grads=(cam.camera_center[None], b.gcol[None]) if fused else None, parts=1)
            b.optimizer_step(grad_scale=0.5, colour_grads=(cam.camera_center[None], b.gcol[None]) if fused else None, parts=2)
            assert a.step_count == b.step_count == it
            assert torch.equal(a.theta, b.theta) and torch.equal(a.m, b.m) and torch.equal(a.v, b.v) and torch.equal(a.act, b.act)


def test_band_loss_shares_add_up_to_the_full_loss():
    """surfel_losses.train_loss_band (tile-band sharding): the band shares computed on band + 32-row halo add up to the full-image
    loss terms, and the band rows of their gradients equal the unsharded gradients (single process: the halo rows are sliced
    from the full tensors instead of received from a neighbour)."""
    import torch
    import surfel_dist as sd
    import surfel_losses as L
    from surfel_render import post_consts_rows
    d = dev()
    g = torch.Generator().manual_seed(3)
    H, W = 200, 144
    img = torch.rand((3, H, W), generator=g).to(d)
    gtimg = (img + 0.1 * torch.randn((3, H, W), generator=g).to(d)).clamp(0, 1)
    am = torch.rand((7, H, W), generator=g).to(d)
    am[1] = am[1] * 0.5 + 0.5; am[0] = am[0] + 2.0 * am[1]; am[5] = 2.0 + torch.rand((H, W), generator=g).to(d)
    cam = torch.zeros(24, device=d)
    cam[0:9] = torch.eye(3, device=d).reshape(-1)
    cam[9:18] = torch.tensor([[1 / 120.0, 0, 0], [0, 1 / 120.0, 0], [-0.6, -0.8, 1.0]], device=d).reshape(-1)
    lam, ln, ld, ratio = 0.2, 0.05, 100.0, 1.0
    x = img.clone().requires_grad_(True); a = am.clone().requires_grad_(True)
    total, sc = L.train_loss(x, a, gtimg, cam, ratio, lam, ln, ld)
    total.backward()
    for world in (2, 3):
        bounds = sd.band_bounds(H, world, multiple=sd.HALO)
        acc_sums = torch.zeros(4, device=d); share_sum = 0.0
        gi = torch.zeros_like(img); ga = torch.zeros_like(am)
        for r, (y0, y1) in enumerate(bounds):
            top, bot = sd.halo_rows(bounds, r, H)
            xe = img[:, y0 - top:y1 + bot].clone().requires_grad_(True); ae = am[:, y0 - top:y1 + bot].clone().requires_grad_(True)
            share, sums = L.train_loss_band(xe, ae, gtimg[:, y0 - top:y1 + bot], post_consts_rows(cam, y0 - top), ratio, lam, ln, ld,
                                            (top, top + y1 - y0), (H, W))
            share.backward()
            gi[:, y0:y1] = xe.grad[:, top:top + y1 - y0]; ga[:, y0:y1] = ae.grad[:, top:top + y1 - y0]
            acc_sums += sums; share_sum += float(share)
        s6 = L.scalars_from_band_sums(acc_sums, float(3 * H * W), float(H * W), lam, ln, ld)
        assert torch.allclose(s6, sc, rtol=2e-5, atol=1e-6), (s6, sc)
        assert abs(share_sum + lam - float(total)) < 1e-5 * max(1.0, abs(float(total)))      # the shares omit the constant lambda_dssim term
        assert torch.allclose(gi, x.grad, rtol=1e-5, atol=1e-9)
        assert torch.allclose(ga, a.grad, rtol=1e-4, atol=1e-9)


def test_compute_cov3D_python_trains_like_the_native_path():
    """pipe.compute_cov3D_python (gaussian_renderer/__init__.py:64-75): the homography built by PyTorch code from the store must
    deliver the same xyz / scaling / rotation gradients into the gradient store as the kernel's own homography (this
    path used to freeze scaling and rotation).  Rotation: compared after the optimiser's normalisation Jacobian (tangential part)."""
    import torch
    import surfel_render as R
    import surfel_trainer as TR
    from surfel_losses import train_loss
    d = dev()
    cams = TR.orbit_cameras(2, 112, 96, device=d)
    bg = torch.zeros(3, device=d)
    gtm = TR.synthetic_object(3000, d, seed=4, px_scale=0.07)
    TR.capture_views(gtm, cams, bg)
    cam = cams[1]
    out = {}
    for python_cov in (False, True):
        m = TR.synthetic_object(3000, d, seed=4, px_scale=0.07)
        m._pv["xyz"].add_(0.01); m.refresh_activations()
        m.spatial_lr_scale = 1.0
        m.training_setup(TR.optimization_params())
        m.bind(sh_grad=True)
        m.grad.fill_(float("nan"))
        pipe = TR.pipeline_params(depth_ratio=1.0, compute_cov3D_python=python_cov)
        img, radii, allmap, m2 = R.rasterize(cam, m, pipe, bg)
        # lambda_normal = 0: with a precomputed homography the rasterizer has no surfel normal (it renders (0, 0, 1) alpha, as the
        # reference's precomp branch does), so only the photometric and distortion terms are comparable between the two paths
        loss, _ = train_loss(img, allmap, cam.original_image, cam.post_consts(), 1.0, 0.2, 0.0, 10.0)
        loss.backward()
        assert torch.isfinite(m.grad).all()
        q = m._av["rotation"]
        grot = m._gv["rotation"] - (m._gv["rotation"] * q).sum(1, keepdim=True) * q
        out[python_cov] = dict(xyz=m._gv["xyz"].clone(), scaling=m._gv["scaling"].clone(), rotation=grot.clone(), opacity=m._gv["opacity"].clone(),
                               sh=m._gv["sh"].clone())
    for k in out[False]:
        a, b = out[False][k].cpu().numpy(), out[True][k].cpu().numpy()
        assert np.abs(a).max() > 0
        assert cosine(a, b) > 0.9999, (k, cosine(a, b))


def test_ssim_per_image_matches_torch():
    """ssim(..., size_average=False) on [B,C,H,W] (utils/loss_utils.py:70-73) against a plain PyTorch fp32 statement of the
    reference's _ssim (grouped 11x11 Gaussian conv, zero padding), value and gradient."""
    import torch
    import torch.nn.functional as F
    import surfel_losses as L
    d = dev()
    g = torch.Generator().manual_seed(1)
    B, C, H, W = 3, 3, 70, 90
    a = torch.rand((B, C, H, W), generator=g).to(d).requires_grad_(True)
    b = (a.detach() + 0.1 * torch.randn((B, C, H, W), generator=g).to(d)).clamp(0, 1)
    got = L.ssim(a, b, size_average=False)
    wts = torch.randn(B, generator=g).to(d)
    (got * wts).sum().backward()
    ga = a.grad.clone(); a.grad = None
    k1 = torch.tensor([math.exp(-(x - 5) ** 2 / (2 * 1.5 ** 2)) for x in range(11)], device=d); k1 = k1 / k1.sum()
    win = (k1[:, None] * k1[None, :]).expand(C, 1, 11, 11).contiguous()
    conv = lambda t: F.conv2d(t, win, padding=5, groups=C)
    mu1, mu2 = conv(a), conv(b)
    s1, s2, s12 = conv(a * a) - mu1 * mu1, conv(b * b) - mu2 * mu2, conv(a * b) - mu1 * mu2
    C1, C2 = 0.01 ** 2, 0.03 ** 2
    ref = (((2 * mu1 * mu2 + C1) * (2 * s12 + C2)) / ((mu1 * mu1 + mu2 * mu2 + C1) * (s1 + s2 + C2))).mean(1).mean(1).mean(1)
    (ref * wts).sum().backward()
    assert got.shape == (B,) and torch.allclose(got, ref, rtol=1e-4, atol=2e-5)
    grad_ok(ga.cpu().numpy(), a.grad.cpu().numpy(), frac=0.999, cos=0.9999)


@pytest.mark.parametrize("ws", [3, 5, 7, 9, 13, 15])
def test_ssim_other_window_sizes_match_torch(ws):
    """ssim(window_size=ws) (utils/loss_utils.py:43: the reference's argument, default 11): the kernels are templates on the window
    radius — value and gradient against a plain PyTorch fp32 statement of the reference's _ssim, both reductions; even or
    out-of-range sizes raise."""
    import torch
    import torch.nn.functional as F
    import surfel_losses as L
    d = dev()
    g = torch.Generator().manual_seed(ws)
    B, C, H, W = 2, 3, 67, 101
    a = torch.rand((B, C, H, W), generator=g).to(d).requires_grad_(True)
    b = (a.detach() + 0.1 * torch.randn((B, C, H, W), generator=g).to(d)).clamp(0, 1)
    k1 = torch.tensor([math.exp(-(x - ws // 2) ** 2 / (2 * 1.5 ** 2)) for x in range(ws)], device=d); k1 = k1 / k1.sum()
    win = (k1[:, None] * k1[None, :]).expand(C, 1, ws, ws).contiguous()
    conv = lambda t: F.conv2d(t, win, padding=ws // 2, groups=C)

    def ref_map(x, y):
        mu1, mu2 = conv(x), conv(y)
        s1, s2, s12 = conv(x * x) - mu1 * mu1, conv(y * y) - mu2 * mu2, conv(x * y) - mu1 * mu2
        C1, C2 = 0.01 ** 2, 0.03 ** 2
        return ((2 * mu1 * mu2 + C1) * (2 * s12 + C2)) / ((mu1 * mu1 + mu2 * mu2 + C1) * (s1 + s2 + C2))

    got = L.ssim(a, b, window_size=ws)                      # size_average=True: one scalar
    got.backward(); ga = a.grad.clone(); a.grad = None
    ref = ref_map(a, b).mean(); ref.backward(); gr = a.grad.clone(); a.grad = None
    assert abs(float(got) - float(ref)) < 2e-5
    grad_ok(ga.cpu().numpy(), gr.cpu().numpy(), frac=0.999, cos=0.9999)
    wts = torch.randn(B, generator=g).to(d)
    got = L.ssim(a, b, window_size=ws, size_average=False)
    (got * wts).sum().backward(); ga = a.grad.clone(); a.grad = None
    ref = ref_map(a, b).mean(1).mean(1).mean(1)
    (ref * wts).sum().backward()
    assert torch.allclose(got, ref, rtol=1e-4, atol=2e-5)
    grad_ok(ga.cpu().numpy(), a.grad.cpu().numpy(), frac=0.999, cos=0.9999)
    for bad in (2, 10, 17, 1):
        with pytest.raises(NotImplementedError):
            L.ssim(a, b, window_size=bad)


def test_sh_degree_below_three(tmp_path):
    """--sh_degree 0..2 (/root/reference/arguments/__init__.py:49): the store keeps 16 coefficients, the inactive ones stay zero
    through training, and .ply / checkpoint hold (d+1)^2 - 1 f_rest coefficients like the reference's."""
    import torch
    import surfel_model
    import surfel_trainer as TR
    d = dev()
    for deg in (0, 1, 2):
        nc = (deg + 1) ** 2
        rng = np.random.default_rng(deg)
        pcd = type("PCD", (), {})(); pcd.points = rng.normal(size=(1500, 3)).astype(np.float32) * 0.6; pcd.colors = rng.random((1500, 3)).astype(np.float32)
        m = surfel_model.GaussianModel(deg, device=d)
        m.create_from_pcd(pcd, spatial_lr_scale=1.0)
        assert m._features_rest.shape == (1500, nc - 1, 3)
        gtm = TR.synthetic_object(1500, d, seed=1, px_scale=0.08)
        cams = TR.capture_views(gtm, TR.orbit_cameras(3, 64, 48, device=d), torch.zeros(3, device=d))
        tr = TR.Trainer(m, cams, TR.optimization_params(sh_degree_interval=2, densify_from_iter=10 ** 6), TR.pipeline_params())
        for _ in range(8):
            tr.step()
        assert m.active_sh_degree == deg                                   # oneupSHdegree stops at max_sh_degree
        sh = m._pv["sh"].view(m.P, 16, 3)
        assert float(sh[:, nc:].abs().max()) == 0.0 if nc < 16 else True   # inactive coefficients never move
        if deg > 0:
            assert float(sh[:, 1:nc].abs().max()) > 0.0                    # active ones train
        path = str(tmp_path / ("pc%d.ply" % deg))
        m.save_ply(path)
        m2 = surfel_model.GaussianModel(deg, device=d); m2.load_ply(path)
        assert torch.equal(m2._pv["sh"], m._pv["sh"]) and torch.equal(m2._xyz, m._xyz)
        cap = m.capture()
        assert cap[3].shape == (m.P, nc - 1, 3)
        m3 = surfel_model.GaussianModel(deg, device=d); m3.restore(cap, TR.optimization_params())
        assert torch.equal(m3.theta, m.theta) and torch.allclose(m3.m, m.m)


def test_view_parallel_step_rehearsed_on_rccl_with_early_gather():
    """The N > 1 view-parallel step run against the real backend (RCCL, world_size 1) on this GPU: view schedule, asynchronous
    all-gather / all-reduce with stream-level waits, Adam in two parts.  With `early_gather` the colour-gradient all-gather is
    launched from INSIDE the rasterizer's backward (surfel_set_backward_hook) while the geometry chain rule still runs; it must
    take exactly the steps of the gather-after-backward form.  (Halo p2p and > 1 rank: gloo tests in tests/test_dist_cpu.py.)"""
    import torch
    import torch.distributed as dist
    import surfel_trainer as TR
    d = dev()
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29541")
    assert not dist.is_initialized()
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=d)
    try:
        bg = torch.zeros(3, device=d)
        gt_model = TR.synthetic_object(4000, d, seed=1, px_scale=0.06)
        cams = TR.capture_views(gt_model, TR.orbit_cameras(4, 128, 96, device=d), bg)
        thetas = []
        for early in (False, True):
            m = TR.synthetic_object(4000, d, seed=2, px_scale=0.05)
            m.spatial_lr_scale = 1.0
            tr = TR.Trainer(m, cams, TR.optimization_params(dist_from_iter=0, normal_from_iter=0, lambda_dist=10.0, densify_from_iter=10 ** 9),
                            TR.pipeline_params(depth_ratio=1.0), rehearse_exchange=True)
            assert tr._exchange and tr._async_exchange and tr.fused_sh
            tr.early_gather = early
            calls = []
            if early:
                inner = tr._on_colour_ready
                tr._on_colour_ready = lambda: (calls.append(1), inner())[1]
            for _ in range(4):
                tr.step()
            torch.cuda.synchronize()
            assert torch.isfinite(tr.last["scalars"]).all() and torch.isfinite(m.theta).all()
            assert len(calls) == (4 if early else 0) and tr._early is None
            thetas.append((m.theta.clone(), m.m.clone(), m.v.clone()))
        for a, b in zip(*thetas):
            assert torch.equal(a, b)
        # early_gather = "auto" (the default with more than one rank): the start-up probe runs EG_LEN iterations with the early
        # gather, EG_LEN without, reduces the two device times over the ranks and fixes the choice — same bits whatever it picks
        m = TR.synthetic_object(4000, d, seed=2, px_scale=0.05)
        m.spatial_lr_scale = 1.0
        tr = TR.Trainer(m, cams, TR.optimization_params(dist_from_iter=0, normal_from_iter=0, lambda_dist=10.0, densify_from_iter=10 ** 9),
                        TR.pipeline_params(depth_ratio=1.0), rehearse_exchange=True)
        tr.early_gather = "auto"
        for _ in range(TR.Trainer.EG_WARM + 2 * TR.Trainer.EG_LEN + 2):
            tr.step()
            if tr.iteration == 4:
                ref4 = (m.theta.clone(), m.m.clone(), m.v.clone())
        torch.cuda.synchronize()
        assert tr.early_gather in (True, False) and tr.early_gather_probe["choice"] in ("early", "late"), tr.early_gather_probe
        assert tr.early_gather_probe["early_ms_per_step"] > 0 and tr.early_gather_probe["late_ms_per_step"] > 0
        for a, b in zip(ref4, thetas[0]):
            assert torch.equal(a, b)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("ratio,white,lam_dist", [(0.3, False, 100.0), (1.0, True, 1000.0), (1.0, False, 1000.0)])
def test_whole_training_iterations_match_the_oracle_chain(ratio, white, lam_dist):
    """Three complete training iterations (train.py:54-138) of the HIP trainer against the same iterations composed from the CPU
    oracles, every stage in fp64: oracle rasterizer forward -> L1 + SSIM and allmap post-processing + regularisers
    (oracle/train_oracle.py, pinned to the reference's own Python) -> oracle rasterizer backward -> the reference's Adam set-up
    on the reference's activations (AdamOracle) with the position-lr schedule.  Iteration k's loss terms depend on every update
    before it, so matching scalars over the three iterations checks the glue between the individually tested stages (gradient
    routing, lambda scaling, activations, learning rates).  Parameters: Adam's first steps move every element by ~lr * sign(g), so
    an element whose gradient sits in fp32 noise may land 2 lr away — a fraction-close bar, as for the gradients."""
    import torch
    import surfel_model
    import surfel_trainer as TR
    from oracle import train_oracle as T
    from oracle.surfel_oracle import Oracle
    d = dev()
    W, H, P = 80, 64, 600
    cams = TR.orbit_cameras(1, W, H, device=d)
    bg = torch.ones(3, device=d) if white else torch.zeros(3, device=d)
    gt_model = TR.synthetic_object(P, d, seed=4, px_scale=0.08)
    TR.capture_views(gt_model, cams, bg)
    cam = cams[0]
    m = TR.synthetic_object(P, d, seed=5, px_scale=0.07)
    m.spatial_lr_scale = 2.0
    raw0 = {k: m._pv[k].detach().cpu().numpy().astype(np.float64).copy() for k in ("xyz", "opacity", "scaling", "rotation")}
    sh0 = m._pv["sh"].detach().cpu().numpy().astype(np.float64).reshape(P, 16, 3).copy()
    # (lam_dist = 1000, depth_ratio = 1: the reference's DTU configuration, /root/reference/scripts/dtu_eval.py:23)
    opt = TR.optimization_params(lambda_dist=lam_dist, lambda_normal=0.05, dist_from_iter=0, normal_from_iter=0, densify_from_iter=10 ** 9,
                                 opacity_reset_interval=10 ** 9)
    tr = TR.Trainer(m, cams, opt, TR.pipeline_params(depth_ratio=ratio), white_background=white)
    hip_scalars = []
    for _ in range(3):
        tr.step()
        hip_scalars.append(tr.last["scalars"].detach().cpu().numpy().astype(np.float64))
    torch.cuda.synchronize()

    # ---- the same three iterations on the CPU, fp64
    O = Oracle("f64")
    A = T.AdamOracle(raw0["xyz"], sh0[:, :1], sh0[:, 1:], raw0["opacity"], raw0["scaling"], raw0["rotation"])
    gt = cam.original_image.cpu().numpy().astype(np.float64)
    wvt = cam.world_view_transform.cpu().numpy(); fpt = cam.full_proj_transform.cpu().numpy(); campos = cam.camera_center.cpu().numpy()
    tanx, tany = math.tan(cam.FoVx * 0.5), math.tan(cam.FoVy * 0.5)
    p = {k: v.detach().numpy() for k, v in A.p.items()}
    ora_scalars = []
    for it in (1, 2, 3):
        o, s, r = T.activate(p["opacity"], p["scaling"], p["rotation"])
        feats = np.concatenate([p["f_dc"], p["f_rest"]], axis=1)
        R, col, oth, radii, st = O.rasterize_forward(np.ones(3) if white else np.zeros(3), p["xyz"], None, o, s, r, 1.0, None, wvt, fpt, tanx, tany, H, W, feats, 3, campos)
        ph = T.photometric(col, gt, opt.lambda_dssim)
        rp = T.render_post_np(oth, wvt, fpt, W, H, ratio, lambda_normal=opt.lambda_normal, lambda_dist=opt.lambda_dist)
        total = ph["loss"] + opt.lambda_normal * rp["normal_err_mean"] + opt.lambda_dist * rp["dist_mean"]
        ora_scalars.append(np.array([ph["l1"], ph["ssim"], rp["normal_err_mean"], rp["dist_mean"], ph["loss"], total]))
        g = O.rasterize_backward(st, ph["g_loss"], np.nan_to_num(rp["g_reg"], nan=0.0))
        lr_xyz = T.expon_lr(it, opt.position_lr_init * 2.0, opt.position_lr_final * 2.0, lr_delay_mult=opt.position_lr_delay_mult,
                            max_steps=opt.position_lr_max_steps)
        lrs = [lr_xyz, opt.feature_lr, opt.feature_lr / 20.0, opt.opacity_lr, opt.scaling_lr, opt.rotation_lr]
        p = A.step(lrs, g.dL_dmeans3D, g.dL_dsh, g.dL_dopacity, g.dL_dscales, g.dL_drots)
    for k in range(3):
        rel = np.abs(hip_scalars[k] - ora_scalars[k]) / np.maximum(np.abs(ora_scalars[k]), 1e-6)
        print("iteration %d: max relative deviation of the six loss scalars %.2e" % (k + 1, rel.max()))
        assert (rel < 1e-4).all(), (k, hip_scalars[k], ora_scalars[k])        # measured 6e-6 .. 3e-5
    # parameters after three steps (synthetic_object starts at SH degree 3: all 16 coefficients are live)
    hip = {k: m._pv[k].detach().cpu().numpy().astype(np.float64) for k in ("xyz", "opacity", "scaling", "rotation")}
    hip_sh = m._pv["sh"].detach().cpu().numpy().astype(np.float64).reshape(P, 16, 3)
    lr_of = dict(xyz=opt.position_lr_init * 2.0, opacity=opt.opacity_lr, scaling=opt.scaling_lr, rotation=opt.rotation_lr)
    for k in hip:
        moved = np.abs(p[k] - raw0[k]).max()
        assert moved > 0.5 * lr_of[k], k                                   # the parameters did move
        frac = (np.abs(hip[k] - p[k]) <= 0.05 * lr_of[k] + 1e-6 * np.abs(p[k])).mean()
        print("%s: %.4f of the elements within 5 %% of a learning rate after three steps" % (k, frac))
        assert frac >= 0.995, (k, frac)        # measured 1.0000
    frac_dc = (np.abs(hip_sh[:, 0] - p["f_dc"][:, 0]) <= 0.05 * opt.feature_lr).mean()
    frac_rest = (np.abs(hip_sh[:, 1:] - p["f_rest"]) <= 0.05 * opt.feature_lr / 20.0).mean()
    assert frac_dc >= 0.995 and frac_rest >= 0.995, (frac_dc, frac_rest)
    assert np.abs(p["f_rest"] - sh0[:, 1:]).max() > 0.5 * opt.feature_lr / 20.0


def test_view_parallel_semantics_match_the_oracle_chain():
    """The optimisation semantics of N view-parallel ranks (one Adam step on the AVERAGE of the views' gradients, SH gradients
    rebuilt from the N colour gradients and view directions — Trainer.views_per_step, the single-process form of the N-rank
    step) against the fp64 oracle chain: both views' gradients from the oracle rasterizer / losses, averaged, one reference Adam
    step; three iterations, two views."""
    import torch
    import surfel_trainer as TR
    from oracle import train_oracle as T
    from oracle.surfel_oracle import Oracle
    d = dev()
    W, H, P = 72, 56, 500
    cams = TR.orbit_cameras(2, W, H, device=d)
    bg = torch.zeros(3, device=d)
    TR.capture_views(TR.synthetic_object(P, d, seed=14, px_scale=0.08), cams, bg)
    m = TR.synthetic_object(P, d, seed=15, px_scale=0.07)
    m.spatial_lr_scale = 1.5
    raw0 = {k: m._pv[k].detach().cpu().numpy().astype(np.float64).copy() for k in ("xyz", "opacity", "scaling", "rotation")}
    sh0 = m._pv["sh"].detach().cpu().numpy().astype(np.float64).reshape(P, 16, 3).copy()
    opt = TR.optimization_params(lambda_dist=100.0, lambda_normal=0.05, dist_from_iter=0, normal_from_iter=0, densify_from_iter=10 ** 9,
                                 opacity_reset_interval=10 ** 9)
    tr = TR.Trainer(m, cams, opt, TR.pipeline_params(depth_ratio=1.0))
    tr.views_per_step = 2
    last = []
    for _ in range(3):
        tr.step()
        last.append(tr.last["scalars"].detach().cpu().numpy().astype(np.float64))
    torch.cuda.synchronize()

    O = Oracle("f64")
    A = T.AdamOracle(raw0["xyz"], sh0[:, :1], sh0[:, 1:], raw0["opacity"], raw0["scaling"], raw0["rotation"])
    p = {k: v.detach().numpy() for k, v in A.p.items()}
    for it in (1, 2, 3):
        o, s, r = T.activate(p["opacity"], p["scaling"], p["rotation"])
        feats = np.concatenate([p["f_dc"], p["f_rest"]], axis=1)
        acc, totals = None, []
        for cam in cams:
            wvt = cam.world_view_transform.cpu().numpy(); fpt = cam.full_proj_transform.cpu().numpy()
            R, col, oth, radii, st = O.rasterize_forward(np.zeros(3), p["xyz"], None, o, s, r, 1.0, None, wvt, fpt, math.tan(cam.FoVx * 0.5),
                                                         math.tan(cam.FoVy * 0.5), H, W, feats, 3, cam.camera_center.cpu().numpy())
            ph = T.photometric(col, cam.original_image.cpu().numpy().astype(np.float64), opt.lambda_dssim)
            rp = T.render_post_np(oth, wvt, fpt, W, H, 1.0, lambda_normal=opt.lambda_normal, lambda_dist=opt.lambda_dist)
            totals.append(ph["loss"] + opt.lambda_normal * rp["normal_err_mean"] + opt.lambda_dist * rp["dist_mean"])
            g = O.rasterize_backward(st, ph["g_loss"], np.nan_to_num(rp["g_reg"], nan=0.0))
            gs = [g.dL_dmeans3D, g.dL_dsh, g.dL_dopacity, g.dL_dscales, g.dL_drots]
            acc = gs if acc is None else [a + b for a, b in zip(acc, gs)]
        # the trainer reports the scalars of the step's last view: one of the two
        assert min(abs(last[it - 1][5] - t) / t for t in totals) < 1e-4, (last[it - 1][5], totals)
        lr_xyz = T.expon_lr(it, opt.position_lr_init * 1.5, opt.position_lr_final * 1.5, lr_delay_mult=opt.position_lr_delay_mult,
                            max_steps=opt.position_lr_max_steps)
        p = A.step([lr_xyz, opt.feature_lr, opt.feature_lr / 20.0, opt.opacity_lr, opt.scaling_lr, opt.rotation_lr], *[0.5 * a for a in acc])
    hip = {k: m._pv[k].detach().cpu().numpy().astype(np.float64) for k in ("xyz", "opacity", "scaling", "rotation")}
    hip_sh = m._pv["sh"].detach().cpu().numpy().astype(np.float64).reshape(P, 16, 3)
    lr_of = dict(xyz=opt.position_lr_init * 1.5, opacity=opt.opacity_lr, scaling=opt.scaling_lr, rotation=opt.rotation_lr)
    for k in hip:
        frac = (np.abs(hip[k] - p[k]) <= 0.05 * lr_of[k] + 1e-6 * np.abs(p[k])).mean()
        print("%s: %.4f of the elements within 5 %% of a learning rate" % (k, frac))
        assert frac >= 0.995, (k, frac)
    frac_dc = (np.abs(hip_sh[:, 0] - p["f_dc"][:, 0]) <= 0.05 * opt.feature_lr).mean()
    frac_rest = (np.abs(hip_sh[:, 1:] - p["f_rest"]) <= 0.05 * opt.feature_lr / 20.0).mean()
    print("sh: dc %.4f rest %.4f" % (frac_dc, frac_rest))
    assert frac_dc >= 0.995 and frac_rest >= 0.995, (frac_dc, frac_rest)
