"""-m gpu parity tests: libsurfel_hip.so (through its C ABI) against the fp64 CPU oracle.

Stated fp32 tolerance (device fp32 vs oracle fp64), the same in every test of this file:
  element test  : images |d| <= 1e-4 + 1e-4*|ref| ; grads |d| <= 1e-4*mean|ref| + 2e-3*|ref|
  bar           : >= 99.9 % of the fp32-DETERMINED elements pass (C5: >= 99.5 %), cosine >= 0.9999 on them, radii exact wherever
                  fp32 pins the extent's ceil() — SURVEY.md 8d's formulation: "threshold-crossing pixels exempt, count reported".
Which elements are exempt is not fitted to the device's results: tests/determinacy.py asks the ORACLE — N Monte-Carlo-arithmetic
evaluations of the same algorithm (every operation with a random fp32-sized rounding error) next to the fp64 one; an element is
determined if all of them agree with fp64 to within half the tolerance.  The exempt elements are counted per tensor and split
into "a decision flipped in some draw" (alpha < 1/255, T(1-alpha) < 1e-4, rho3d <= rho2d, depth < 0.2, T > 0.5, ceil(extent), SH
clamp — by the oracle's decision signatures) and "same decisions, ill-conditioned sums" (k = px*Tw - Tu cancels to ~1e-4 relative;
cx^2 - sum(f*Tu*Tu) in the AABB).  Small scenes are additionally held on ALL elements (>= 99.9 % or <= 3 surfels off).  As a second
yardstick the config-size tests print the same algorithm run in plain fp32 on the CPU (oracle -DORACLE_F32) and require the
device's all-element fraction to be no more than 0.2 % below it.
The device's float32 view depths are injected into every oracle run as the sort key so all sides order near-ties identically.
"""
import numpy as np
import pytest

import determinacy as D
from helpers import HipRun, cosine, frac_close, oracle_forward, scene_args

pytestmark = pytest.mark.gpu

IMG_ATOL, IMG_RTOL, IMG_FRAC = D.IMG_ATOL, D.IMG_RTOL, D.PASS_FRAC
LARGE_SORT_DEFAULT = 2          # surfel_set_option("large_sort") default of the library (profiles/r02_large_sort.md)
G_RTOL, G_FRAC, G_COS = D.G_RTOL, D.PASS_FRAC, D.COS_MIN
GOLDEN_IMG_FRAC, GOLDEN_G_FRAC, GOLDEN_G_COS = 0.9995, 0.999, 0.99999      # test_golden_fixture (512 surfels, 64x48): all elements, no exemption
C5_PASS_FRAC = 0.995        # SURVEY 8d's bar at the 10 M-surfel stress size (2 draws there: fewer exempt elements found, a looser bar on the rest)
STRESS_EXEMPT_CAP = 0.6     # scenes BUILT to be ill-conditioned (distortion gradient x 30 000, discs of hundreds of pixels): the probe may exempt more


def _scene(name_or_dims, seed=0, **kw):
    import synthetic
    if isinstance(name_or_dims, str):
        sc = synthetic.make_config(name_or_dims, seed=seed)
    else:
        P, W, H = name_or_dims
        sc = synthetic.make_scene(P, W, H, seed=seed, **kw)
    return sc


def _check_binning(run, R, radii):
    """Small scenes: radii exact.  The device emits a (tile, surfel) instance only where the surfel's alpha >= 1/255 bbox reaches the
    tile, so its instance count is a subset of the reference rect count the oracle reports.  (Config sizes: determinacy.judge_radii.)"""
    got = run.radii.cpu().numpy()
    assert np.array_equal(got, radii) and run.R <= R, (int((got != radii).sum()), run.R, R)


def _check_images(run, col, oth, st):
    c = run.color.cpu().numpy(); o = run.others.cpu().numpy()
    assert np.isfinite(c).all() and np.isfinite(o).all()
    assert frac_close(c, col, IMG_ATOL, IMG_RTOL) >= IMG_FRAC, "color"
    for ch, nm in [(0, "depth-sum"), (1, "alpha"), (2, "nx"), (3, "ny"), (4, "nz"), (6, "distortion")]:
        f = frac_close(o[ch], oth[ch], IMG_ATOL, IMG_RTOL)
        assert f >= IMG_FRAC, "%s: only %.5f of pixels within tolerance" % (nm, f)
    assert frac_close(o[5], oth[5], 1e-4, 1e-4) >= IMG_FRAC, "median depth"


def _check_grads(g, og, has_sr=True):
    pairs = [("means3D", og.dL_dmeans3D), ("opacity", og.dL_dopacity), ("sh", og.dL_dsh), ("means2D", og.dL_dmean2D)]
    if has_sr:
        pairs += [("scales", og.dL_dscales), ("rots", og.dL_drots)]
    for k, ref in pairs:
        x = g[k].reshape(ref.shape)
        assert np.isfinite(x).all(), k
        scale = np.abs(ref).mean() + 1e-30
        f = frac_close(x, ref, 1e-4 * scale + 1e-12, G_RTOL)
        cs = cosine(x, ref)
        # a (pixel, surfel) pair sitting exactly on the 1/255 or 1e-4 threshold may be decided differently in fp32;
        # that moves ONE surfel's gradient, so small scenes get an absolute allowance of 3 surfels
        bad = np.abs(x.astype(np.float64) - ref) > (1e-4 * scale + 1e-12 + G_RTOL * np.abs(ref))
        bad_surfels = int(bad.reshape(bad.shape[0], -1).any(1).sum())
        assert f >= G_FRAC or bad_surfels <= 3, "%s: only %.5f of elements within tolerance, %d surfels (cos %.7f)" % (k, f, bad_surfels, cs)
        assert cs >= G_COS, "%s: cosine %.7f" % (k, cs)


@pytest.mark.parametrize("dims,kw", [((512, 72, 56), dict(seed=7, px_radius=4.0, z_near=1.0, z_far=6.0)),
                                     ((3000, 200, 120), dict(seed=2, px_radius=6.0, z_near=0.5, z_far=8.0)),
                                     ((50, 33, 17), dict(seed=3, px_radius=3.0))])
def test_forward_backward_small(dims, kw):
    from oracle.surfel_oracle import Oracle
    sc = _scene(dims, **kw)
    sc["bg"] = np.array([0.2, 0.5, 0.9], np.float32)
    a = scene_args(sc)
    run = HipRun(a).forward()
    o = Oracle("f64")
    R, col, oth, radii, st = oracle_forward(o, a, depth_key=run.depths())
    _check_binning(run, R, radii)
    _check_images(run, col, oth, st)
    rng = np.random.default_rng(5)
    gC = rng.normal(size=col.shape).astype(np.float32); gO = rng.normal(size=oth.shape).astype(np.float32)
    g = run.backward(gC, gO)
    og = o.rasterize_backward(st, gC, gO)
    _check_grads(g, og)


def test_golden_fixture(golden):
    """Committed fixture: inputs captured from the reference's own render() call + fp64 oracle outputs."""
    a = dict(bg=golden["bg"], means3D=golden["means3D"], opacities=golden["opacities"], scales=golden["scales"],
             rotations=golden["rotations"], shs=golden["shs"], viewmatrix=golden["viewmatrix"], projmatrix=golden["projmatrix"],
             campos=golden["campos"], tanfovx=float(golden["tanfovx"]), tanfovy=float(golden["tanfovy"]),
             W=int(golden["image_width"]), H=int(golden["image_height"]), sh_degree=int(golden["sh_degree"]), scale_modifier=1.0)
    run = HipRun(a).forward()
    assert 0 < run.R <= int(golden["oracle_R"])
    assert np.array_equal(run.radii.cpu().numpy(), golden["oracle_radii"])
    fc = frac_close(run.color.cpu().numpy(), golden["oracle_color"], IMG_ATOL, IMG_RTOL)
    fo = frac_close(run.others.cpu().numpy()[:5], golden["oracle_others"][:5], IMG_ATOL, IMG_RTOL)
    print("golden: pixel frac colour %.5f, allmap[0:5] %.5f" % (fc, fo))
    assert fc >= GOLDEN_IMG_FRAC and fo >= GOLDEN_IMG_FRAC
    g = run.backward(golden["grad_color"], golden["grad_others"])
    for k, ref in [("means3D", golden["oracle_dL_dmeans3D"]), ("scales", golden["oracle_dL_dscales"]),
                   ("rots", golden["oracle_dL_drots"]), ("opacity", golden["oracle_dL_dopacity"]), ("sh", golden["oracle_dL_dsh"])]:
        x = g[k].reshape(ref.shape)
        fr, cs = frac_close(x, ref, 1e-4 * np.abs(ref).mean(), G_RTOL), cosine(x, ref)
        print("golden dL/d%s: frac %.5f cosine %.7f" % (k, fr, cs))
        assert fr >= GOLDEN_G_FRAC and cs >= GOLDEN_G_COS, "%s: frac %.5f cosine %.7f" % (k, fr, cs)
    # the reference's own in-tree transMat (gaussian_renderer/__init__.py:64-75) fed as cov3D_precomp must
    # render the same colour image as the native scale/rotation path (SURVEY.md §4 self-consistency)
    # ... at the stated image tolerance, on the pixels fp32 determines, colour AND depth / alpha / median depth / distortion (the precomp
    # path carries no normal: channels 2-4 differ by design): the reference's T (torch fp32) and the kernel's T differ by roundings
    run2 = HipRun(a, transMat_precomp=golden["ref_cov3D_precomp"]).forward()
    det = D.Determinacy(a, run.depths(), n_draws=5)
    D.judge_images("golden native", det, run.color.cpu().numpy(), run.others.cpu().numpy())
    m = det.image_masks()
    c2, o2 = run2.color.cpu().numpy(), run2.others.cpu().numpy()
    D.judge("golden cov3D_precomp twin", "color", c2, *m["color"], D.img_tol(m["color"][0]))
    for ch in (0, 1, 5, 6):
        r = m["others%d" % ch]
        D.judge("golden cov3D_precomp twin", "others%d" % ch, o2[ch], *r, D.img_tol(r[0]))


def _check_depths(run, st, radii):
    """The float32 view depths the device sorts on, against the oracle's OWN fp64 depths (they are also injected into the oracle
    as the sort key so near-ties order identically — which alone would let a wrong device depth go unnoticed)."""
    got = run.depths().astype(np.float64)
    vis = (radii > 0) & (run.radii.cpu().numpy() > 0)
    assert vis.sum() > 0.5 * (radii > 0).sum()
    err = np.abs(got[vis] - st.depths[vis])
    # a 4-term fp32 dot product of O(10)-sized terms: a few ulps of the largest term
    assert err.max() <= 4e-6 and np.median(err / st.depths[vis]) <= 1.2e-7, (err.max(), np.median(err / st.depths[vis]))


CONFIG_DRAWS = {"C1": 5, "C2": 5, "C3": 5, "C4": 4}


def _yardstick(tag, name, x, ref, r32, tol):
    """All elements (no exemption) next to the same algorithm in plain fp32 on the CPU: the device is never worse by more than 0.2 %."""
    x = np.asarray(x, np.float64).reshape(ref.shape)
    f = float((np.abs(x - ref) <= tol).mean()); f32 = float((np.abs(np.asarray(r32, np.float64) - ref) <= tol).mean())
    print("%s %-8s: all elements hip %.5f cpu-fp32 %.5f" % (tag, name, f, f32))
    assert f >= f32 - 0.002, "%s %s: hip %.5f, cpu-fp32 %.5f" % (tag, name, f, f32)


@pytest.mark.parametrize("name", ["C1", "C2", "C3", "C4"])
def test_config_sizes(name):
    """BASELINE configs 1-4 shapes (10k/256^2, 300k/800^2, 200k/800x600 = DTU -r 2 /root/reference/scripts/dtu_eval.py:23,
    2M/1600x1060 = the width-1600 cap of /root/reference/utils/camera_utils.py:25-33) against the fp64 oracle under the module's bar:
    >= 99.9 % of the fp32-determined elements, cosine >= 0.9999, radii exact where pinned, exempt fractions printed (determinacy.py);
    the fp32 CPU run of the same algorithm as the all-element yardstick.  At C3 / C4 both binning paths are forced and must agree
    bit for bit (C4 takes the > 1 M-item three-launch sort)."""
    import surfel_native as n
    from oracle.surfel_oracle import Oracle
    sc = _scene(name)
    a = scene_args(sc)
    run = HipRun(a).forward()
    dk = run.depths()
    det = D.Determinacy(a, dk, n_draws=CONFIG_DRAWS[name])
    R, col, oth, radii, st = det.R, det.col, det.oth, det.radii, det.st
    o32 = Oracle("f32")
    R32, col32, oth32, radii32, st32 = oracle_forward(o32, a, depth_key=dk)
    D.judge_radii(name, det, run.radii.cpu().numpy(), run.R)
    _check_depths(run, st, radii)
    c = run.color.cpu().numpy(); o = run.others.cpu().numpy()
    D.judge_images(name, det, c, o)
    for nm, x, x32, ref in [("color", c, col32, col)] + [("others%d" % i, o[i], oth32[i], oth[i]) for i in range(7)]:
        _yardstick(name, nm, x, ref, x32, D.img_tol(ref))
    rng = np.random.default_rng(9)
    gC = rng.normal(size=col.shape).astype(np.float32); gO = rng.normal(size=oth.shape).astype(np.float32)
    g = run.backward(gC, gO)
    og, og32 = det.backward(gC, gO), o32.rasterize_backward(st32, gC, gO)

    def check(g, walk):
        D.judge_grads("%s %s" % (name, walk), det, g)
        for k, attr in D.GRADS:
            _yardstick("%s %s" % (name, walk), k, g[k], getattr(og, attr), getattr(og32, attr), D.grad_tol(getattr(og, attr)))

    check(g, "auto")
    run.debug = n.OPT_BWD_SCAN          # the scan walk is held to the same bars
    check(run.backward(gC, gO), "scan")
    run.debug = 0
    if name in ("C3", "C4"):
        base = (run.R, c, o, run.radii.cpu().numpy(), g)
        for mode in (0, 2):      # depth-presorted emission | per-tile depth sort
            r2 = HipRun(a, debug=n.opt_tile_sort(mode)).forward()
            g2 = r2.backward(gC, gO)
            assert r2.R == base[0] and np.array_equal(r2.radii.cpu().numpy(), base[3])
            assert np.array_equal(r2.color.cpu().numpy(), base[1]) and np.array_equal(r2.others.cpu().numpy(), base[2]), mode
            for k in g2:
                assert np.array_equal(g2[k], base[4][k]), "%s: dL/d%s differs on binning path %d" % (name, k, mode)


def _trained_state(W, H, iters, lambda_dist, view=3, seed=0):
    """`iters` iterations of the reference schedule (densification from 500; depth_ratio 1, lambda_normal 0.05, the given lambda_dist)
    from 120 k random points on a 24-view synthetic capture -> (numpy arguments of one training view, that view's camera)."""
    import torch
    import surfel_model
    import surfel_trainer as TR
    d = torch.device("cuda:0")
    torch.manual_seed(seed)
    bg = torch.zeros(3, device=d)
    gt = TR.synthetic_object(120_000, d, seed=seed, px_scale=0.035)
    cams = TR.capture_views(gt, TR.orbit_cameras(24, W, H, device=d), bg)
    del gt
    extent = TR.cameras_extent(cams)
    rng = np.random.default_rng(seed)
    pcd = type("PCD", (), {})()
    pcd.points = (rng.random((120_000, 3)) * 2.6 - 1.3).astype(np.float32)
    pcd.colors = rng.random((120_000, 3)).astype(np.float32)
    model = surfel_model.GaussianModel(3, device=d)
    model.create_from_pcd(pcd, spatial_lr_scale=extent)
    tr = TR.Trainer(model, cams, TR.optimization_params(iterations=iters, lambda_dist=lambda_dist, position_lr_max_steps=iters, dist_from_iter=300, normal_from_iter=700),
                    TR.pipeline_params(depth_ratio=1.0), extent=extent)
    for _ in range(iters - 1):
        tr.step()
    torch.cuda.synchronize()
    cam = cams[view]
    f = lambda t: t.detach().float().cpu().numpy()
    a = dict(bg=np.zeros(3, np.float32), means3D=f(model.get_xyz), opacities=f(model.get_opacity), scales=f(model.get_scaling), rotations=f(model.get_rotation),
             shs=f(model.get_features), viewmatrix=f(cam.world_view_transform), projmatrix=f(cam.full_proj_transform), campos=f(cam.camera_center),
             tanfovx=float(np.tan(cam.FoVx * 0.5)), tanfovy=float(np.tan(cam.FoVy * 0.5)), W=W, H=H, sh_degree=int(model.active_sh_degree), scale_modifier=1.0)
    del tr, model
    import diff_surfel_rasterization as dsr
    dsr.set_grad_arena(None)
    return a, cam


def _walk_table(tag, run, gC, gO, det, cap=D.EXEMPT_CAP):
    """Every blend_bwd walk under the module's bar on the fp32-determined elements of `det` (its backward has been run on gC, gO)."""
    import surfel_native as n
    for walk, flag in (("rows", n.OPT_BWD_ROWS), ("quad", n.OPT_BWD_QUAD), ("scan", n.OPT_BWD_SCAN), ("auto", 0)):
        run.debug = flag
        g = run.backward(gC, gO)
        D.judge_grads("%s %s" % (tag, walk), det, g)
    run.debug = 0


def test_trained_state_parity():
    """The regime that dominates every realistic leg: a TRAINED state — wide faint discs, needles, early
    saturation, hundreds of instances on the centre tiles — at 800x800, not random surfels.  1 500 iterations of the reference
    schedule (densification from 500) on a synthetic capture, then one training view of that model: all ten channels and every
    gradient of the rows / quad / scan walks against the fp64 oracle under the module's bar (fp32-determined elements, exempt
    fractions printed)."""
    import torch
    from oracle.surfel_oracle import Oracle
    W = H = 800
    a, cam = _trained_state(W, H, 1500, 100.0)
    P = a["means3D"].shape[0]
    run = HipRun(a).forward()
    dk = run.depths()
    det = D.Determinacy(a, dk, n_draws=4)
    R, col, oth, radii, st = det.R, det.col, det.oth, det.radii, det.st
    gx, gy = (W + 15) // 16, (H + 15) // 16
    ranges = run.ia.last()[:gx * gy * 8].view(torch.int32).view(-1, 2).cpu().numpy()
    lens = ranges[:, 1] - ranges[:, 0]
    print("trained state: %d surfels, R %d (%.1f per surfel), longest tile list %d, empty tiles %d of %d" % (P, run.R, run.R / P, lens.max(), int((lens == 0).sum()), lens.size))
    assert P > 30_000 and run.R > 4 * P and lens.max() > 300      # the regime: many instances per surfel, long centre lists
    D.judge_radii("trained", det, run.radii.cpu().numpy(), run.R)
    _check_depths(run, st, radii)
    D.judge_images("trained", det, run.color.cpu().numpy(), run.others.cpu().numpy())
    rg = np.random.default_rng(9)
    gC = rg.normal(size=col.shape).astype(np.float32); gO = rg.normal(size=oth.shape).astype(np.float32)
    det.backward(gC, gO)
    _walk_table("trained", run, gC, gO, det)


def test_dtu_gradient_regime_parity():
    """the rasterizer backward in the gradient regime of BASELINE configs[2] — a state TRAINED with
    the reference's DTU settings (/root/reference/scripts/dtu_eval.py:23: depth_ratio 1, lambda_dist 1000; lambda_normal 0.05) at the
    DTU -r 2 shape 800x600, and upstream gradients that are the REAL loss's (/root/reference/train.py:77-88: L1 + SSIM on the colour,
    lambda_normal x normal consistency and lambda_dist x distortion through the allmap post-processing), produced by the product's own
    fused loss launch on this frame — not N(0,1) on ten channels.  rows / quad / scan / auto against the fp64 oracle on those same
    upstream gradients under the module's bar; the ratio |g_dist| / |g_colour| is printed."""
    import torch
    import surfel_losses as L
    from oracle.surfel_oracle import Oracle
    W, H = 800, 600
    a, cam = _trained_state(W, H, 1500, 1000.0)
    P = a["means3D"].shape[0]
    run = HipRun(a).forward()
    dk = run.depths()
    det = D.Determinacy(a, dk, n_draws=4)
    R, col, oth, radii, st = det.R, det.col, det.oth, det.radii, det.st
    assert P > 30_000 and run.R > 4 * P
    with torch.no_grad():
        lctx, total, scalars = L.train_loss_manual(run.color, run.others, cam.original_image, cam.post_consts(), 1.0, 0.2, 0.05, 1000.0, defer_scalars=False)
        g_img, g_am = L.train_loss_manual_backward(lctx, torch.ones((), device=run.color.device))
    torch.cuda.synchronize()
    gC, gO = g_img.float().cpu().numpy().reshape(col.shape), g_am.float().cpu().numpy().reshape(oth.shape)
    assert np.isfinite(gC).all() and np.isfinite(gO).all()
    ratio = float(np.abs(gO[6]).mean() / np.abs(gC).mean())
    print("DTU regime: %d surfels, R %d, loss terms %s, mean |g_dist| / mean |g_colour| = %.0f, |g_normal| / |g_colour| = %.2f" %
          (P, run.R, np.round(scalars.cpu().numpy(), 5).tolist(), ratio, float(np.abs(gO[2:5]).mean() / np.abs(gC).mean())))
    assert ratio > 300.0      # the regime: the distortion gradient dominates the colour gradient by orders of magnitude
    det.backward(gC, gO)
    _walk_table("DTU", run, gC, gO, det)


def test_config_c5_stress():
    """BASELINE config 5 (10 M surfels, 3840x2160, ~1e8 tile instances: 32-bit offsets, the > 1 M-item sort on both levels):
    instance count, radii, depths, all ten image channels and every gradient against the fp64 oracle: >= 99.5 % of the
    fp32-determined elements (two Monte-Carlo-arithmetic draws at this size), cosine >= 0.9999."""
    sc = _scene("C5")
    a = scene_args(sc)
    run = HipRun(a).forward()
    det = D.Determinacy(a, run.depths(), n_draws=2)
    assert run.R > 50_000_000
    D.judge_radii("C5", det, run.radii.cpu().numpy(), run.R)
    _check_depths(run, det.st, det.radii)
    D.judge_images("C5", det, run.color.cpu().numpy(), run.others.cpu().numpy(), pass_frac=C5_PASS_FRAC)
    rng = np.random.default_rng(9)
    gC = rng.normal(size=det.col.shape).astype(np.float32); gO = rng.normal(size=det.oth.shape).astype(np.float32)
    g = run.backward(gC, gO)
    det.backward(gC, gO)
    D.judge_grads("C5", det, g, pass_frac=C5_PASS_FRAC)


def test_precomp_and_override_color():
    """cov3D_precomp + colors_precomp branch (render(..., override_color), compute_cov3D_python)."""
    from oracle.surfel_oracle import Oracle
    sc = _scene((800, 96, 80), seed=4, px_radius=5.0)
    a = scene_args(sc)
    o = Oracle("f64")
    _, _, _, _, st0 = oracle_forward(o, a)
    trans = st0.transMat.astype(np.float32)
    cols = np.random.default_rng(1).uniform(0, 1, (a["means3D"].shape[0], 3)).astype(np.float32)
    run = HipRun(a, colors_precomp=cols, transMat_precomp=trans).forward()
    R, col, oth, radii, st = oracle_forward(o, a, colors_precomp=cols, transMat_precomp=trans, depth_key=run.depths())
    assert run.R <= R and np.array_equal(run.radii.cpu().numpy(), radii)
    _check_images(run, col, oth, st)
    rng = np.random.default_rng(2)
    gC = rng.normal(size=col.shape).astype(np.float32); gO = rng.normal(size=oth.shape).astype(np.float32)
    g = run.backward(gC, gO)
    og = o.rasterize_backward(st, gC, gO)
    for k, ref in [("transMat", og.dL_dtransMat), ("colors", og.dL_dcolors), ("opacity", og.dL_dopacity), ("means2D", og.dL_dmean2D)]:
        x = g[k].reshape(ref.shape)
        scale = np.abs(ref).mean() + 1e-30
        assert frac_close(x, ref, 1e-4 * scale, G_RTOL) >= G_FRAC, k
        assert cosine(x, ref) >= G_COS, k


def _stress_scene(kind, seed):
    """Scenes built to break a culling rule: needle-thin, huge, grazing-angle, tiny and near-camera surfels."""
    import synthetic
    rng = np.random.default_rng(seed)
    sc = synthetic.make_scene(1500, 208, 136, seed=seed, px_radius=5.0, z_near=0.6, z_far=9.0)
    P = sc["means3D"].shape[0]
    if kind == "needles":        # 100:1 anisotropy, every orientation
        sc["scales"] = sc["scales"] * np.where(rng.random((P, 1)) < 0.5, [[12.0, 0.12]], [[0.1, 10.0]]).astype(np.float32)
    elif kind == "huge":         # footprints of hundreds of pixels, some crossing the camera plane
        sc["scales"] = sc["scales"] * rng.choice([1.0, 8.0, 40.0], size=(P, 1)).astype(np.float32)
    elif kind == "tiny":         # sub-pixel surfels: the low-pass disc carries the whole footprint
        sc["scales"] = sc["scales"] * rng.choice([0.02, 0.2, 1.0], size=(P, 1)).astype(np.float32)
    elif kind == "grazing":      # discs seen almost edge-on (normal ~ perpendicular to the view ray)
        d = sc["means3D"] - sc["campos"][None]
        d /= np.linalg.norm(d, axis=1, keepdims=True)
        t1 = np.cross(d, rng.normal(size=(P, 3))); t1 /= np.linalg.norm(t1, axis=1, keepdims=True)
        n = t1 + 0.03 * rng.normal(size=(P, 1)) * d                     # normal almost perpendicular to the ray
        n /= np.linalg.norm(n, axis=1, keepdims=True)
        a = np.cross(n, d); a /= np.linalg.norm(a, axis=1, keepdims=True)
        b = np.cross(n, a)
        R = np.stack([a, b, n], 2)                                      # columns = disc axes, normal
        R[np.linalg.det(R) < 0, :, 0] *= -1
        w = np.sqrt(np.maximum(0, 1 + R[:, 0, 0] + R[:, 1, 1] + R[:, 2, 2])) / 2 + 1e-9
        q = np.stack([w, (R[:, 2, 1] - R[:, 1, 2]) / (4 * w), (R[:, 0, 2] - R[:, 2, 0]) / (4 * w), (R[:, 1, 0] - R[:, 0, 1]) / (4 * w)], 1)
        sc["rotations"] = (q / np.linalg.norm(q, axis=1, keepdims=True)).astype(np.float32)
        sc["scales"] = (sc["scales"] * 3.0).astype(np.float32)
    elif kind == "huge_faint":   # wide, nearly transparent discs close to the camera: the 3-sigma box centre runs far off screen
        sc = synthetic.make_scene(4000, 96, 64, seed=seed, px_radius=30.0, z_near=2.0, z_far=8.0)
        sc["opacities"] = np.full_like(sc["opacities"], 0.02)
    elif kind == "opaque_faint":  # opacities at both ends: 1/255-ish (empty footprints) and ~1 (widest footprints)
        sc["opacities"] = rng.choice([0.0035, 0.0045, 0.02, 0.999], size=(P, 1)).astype(np.float32)
    sc["bg"] = np.array([0.1, 0.3, 0.7], np.float32)
    return sc


@pytest.mark.parametrize("kind", ["plain", "needles", "huge", "tiny", "grazing", "opaque_faint", "huge_faint"])
def test_culling_is_exact(kind):
    """Every cull (footprint-restricted tile emission, per-quad / per-sub-tile instance masks) only ever removes
    (pixel, surfel) pairs that contribute nothing: images and gradients must be BIT-IDENTICAL with culling on and off,
    and with culling off the instance count is exactly the reference's rect count."""
    import surfel_native as n
    from oracle.surfel_oracle import Oracle
    lib = n.load()
    o = Oracle("f64")
    for seed in (11, 12):
        a = scene_args(_stress_scene(kind, seed))
        rng = np.random.default_rng(seed)
        gC = rng.normal(size=(3, a["H"], a["W"])).astype(np.float32); gO = rng.normal(size=(7, a["H"], a["W"])).astype(np.float32)
        res = []
        for cull in (1, 0):          # per-call option bit of the debug word: no process-wide switch is touched
            run = HipRun(a, debug=0 if cull else n.OPT_NO_CULL).forward()
            # (the walk is forced: the default picks it from the frame's instances per surfel, which the culling changes, and the
            # walks differ in summation order — the scan walk's even with the list lengths, so only rows / quad can be held to the bits)
            run.debug = n.OPT_BWD_ROWS
            g = run.backward(gC, gO)
            res.append((run.R, run.color.cpu().numpy(), run.others.cpu().numpy(), run.radii.cpu().numpy(), g))
        (R1, c1, o1, r1, g1), (R0, c0, o0, r0, g0) = res
        assert R1 <= R0 and np.array_equal(r1, r0)
        assert np.array_equal(c1, c0), "%s/%d: colour differs with culling" % (kind, seed)
        assert np.array_equal(o1, o0), "%s/%d: allmap differs with culling" % (kind, seed)
        for k in g1:
            assert np.isfinite(g1[k]).all(), "%s/%d: dL/d%s has non-finite entries" % (kind, seed, k)
            assert np.array_equal(g1[k], g0[k]), "%s/%d: dL/d%s differs with culling (max |d| %.3e)" % (
                kind, seed, k, np.abs(g1[k].astype(np.float64) - g0[k]).max())
        # culling off = the reference's binning: instance count equals the oracle's rect count when radii agree
        Rref, _, _, radii, _ = oracle_forward(o, a, depth_key=None)
        if np.array_equal(r0, radii):
            assert R0 == Rref, (R0, Rref)


@pytest.mark.parametrize("kind", ["plain", "huge", "ties", "crowded", "giant"])
def test_binning_paths_are_identical(kind):
    """The two binning paths — depth-presorted emission vs index-order emission + per-tile depth sort in LDS — must give
    BIT-IDENTICAL images and gradients (same per-tile order: depth bits, then surfel index).  'ties' holds many exactly equal
    depths; 'crowded' piles > 4096 instances onto single tiles to take the rank-sort fallback; 'giant' holds discs that cover more
    than 1023 tiles (the depth sort's value carries min(tile count, 1023) above the surfel id: such surfels take the look-up)."""
    import surfel_native as n
    import synthetic
    lib = n.load()
    if kind in ("plain", "huge"):
        sc = _stress_scene(kind, 21)
    elif kind == "giant":
        sc = synthetic.make_scene(2500, 656, 640, seed=8, px_radius=5.0, z_near=2.0, z_far=9.0)      # 41 x 40 = 1640 tiles
        rng0 = np.random.default_rng(8)
        big = rng0.random(sc["means3D"].shape[0]) < 0.004
        sc["scales"][big] *= 30.0
        sc["opacities"][big] = 0.03
    elif kind == "ties":
        sc = synthetic.make_scene(3000, 160, 120, seed=5, px_radius=6.0, z_near=2.0, z_far=6.0, tilt=False)
        # identity rotation camera: view depth = world z + const; quantise z so that hundreds of surfels share a depth bit pattern
        sc["means3D"][:, 2] = np.round(sc["means3D"][:, 2] * 2.0) / 2.0
    else:
        sc = synthetic.make_scene(9000, 96, 64, seed=6, px_radius=30.0, z_near=2.0, z_far=8.0)
        sc["opacities"] = np.full_like(sc["opacities"], 0.02)         # faint: nothing saturates, every instance is blended
    a = scene_args(sc)
    rng = np.random.default_rng(3)
    gC = rng.normal(size=(3, a["H"], a["W"])).astype(np.float32); gO = rng.normal(size=(7, a["H"], a["W"])).astype(np.float32)
    res = []
    for mode in (0, 2, 1, 1):          # presorted | per-tile depth sort | auto (twice: first-frame and steady-state decisions)
        run = HipRun(a, debug=n.opt_tile_sort(mode)).forward()
        g = run.backward(gC, gO)
        res.append((run.R, run.color.cpu().numpy(), run.others.cpu().numpy(), run.radii.cpu().numpy(), g))
    R0, c0, o0, r0, g0 = res[0]
    assert R0 > 0
    if kind == "giant":
        assert (r0 > 400).sum() >= 3, "no surfel large enough to cover 1023 tiles (largest radius %d px)" % r0.max()
    if kind == "crowded":
        tiles = ((a["W"] + 15) // 16) * ((a["H"] + 15) // 16)
        assert R0 > 4096 * tiles * 0.5, "scene not crowded enough to reach the fallback (%d instances on %d tiles)" % (R0, tiles)
    for m, (R2, c2, o2, r2, g2) in enumerate(res[1:]):
        assert R0 == R2 and np.array_equal(r0, r2)
        assert np.array_equal(c0, c2) and np.array_equal(o0, o2), "%s: images differ between the binning paths (%d)" % (kind, m)
        for k in g0:
            assert np.array_equal(g0[k], g2[k]), "%s: dL/d%s differs between the binning paths (%d)" % (kind, k, m)


@pytest.mark.parametrize("kind", ["plain", "needles", "huge", "tiny", "grazing", "opaque_faint", "crowded", "C1"])
def test_backward_variants_are_identical(kind):
    """blend_bwd per-row walk (default) vs per-quad walk (round 1's kernel): same per-pair arithmetic, same summation tree
    ((r0+r1)+(r2+r3) per wave, waves 0..3) -> BIT-IDENTICAL gradients, selected per call through the debug word's option bits.
    'crowded' piles thousands of instances on every tile (many slot rounds per batch, big footprints: 16 slots per instance)."""
    import surfel_native as n
    import synthetic
    if kind == "C1":
        sc = _scene("C1", seed=2)
    elif kind == "crowded":
        sc = synthetic.make_scene(9000, 96, 64, seed=6, px_radius=30.0, z_near=2.0, z_far=8.0)
        sc["opacities"] = np.full_like(sc["opacities"], 0.02)
    else:
        sc = _stress_scene(kind, 31)
    a = scene_args(sc)
    rng = np.random.default_rng(8)
    gC = rng.normal(size=(3, a["H"], a["W"])).astype(np.float32); gO = rng.normal(size=(7, a["H"], a["W"])).astype(np.float32)
    run = HipRun(a).forward()
    res = []
    for flag in (n.OPT_BWD_ROWS, n.OPT_BWD_QUAD, n.OPT_BWD_ROWS):
        run.debug = flag
        res.append(run.backward(gC, gO))
    for k in res[0]:
        assert np.isfinite(res[0][k]).all(), k
        assert np.array_equal(res[0][k], res[2][k]), "rows variant not reproducible: %s" % k
        d = np.abs(res[0][k].astype(np.float64) - res[1][k])
        assert np.array_equal(res[0][k], res[1][k]), "%s: dL/d%s differs between the variants (max |d| %.3e, %d elements)" % (
            kind, k, d.max(), int((d > 0).sum()))


@pytest.mark.parametrize("kind", ["plain", "huge", "crowded", "saturating", "C1"])
def test_tile_cuts_equal_zero_records(kind):
    """Behind a tile's saturation point blend_bwd either writes zero gradient records (small frames) or writes NOTHING and leaves
    the tile's cut — the (depth bits, surfel index) key of its last staged instance — for preprocess_bwd to test before every
    record fetch (frames with >= 2^21 instances).  Both forms, every walk, both record gathers: BIT-IDENTICAL gradients."""
    import surfel_native as n
    a = scene_args(_walk_scene(kind))
    rng = np.random.default_rng(3)
    gC = rng.normal(size=(3, a["H"], a["W"])).astype(np.float32); gO = rng.normal(size=(7, a["H"], a["W"])).astype(np.float32)
    run = HipRun(a).forward()
    for walk in (n.OPT_BWD_ROWS, n.OPT_BWD_QUAD, n.OPT_BWD_SCAN):
        for gather in (n.OPT_PBWD_THREAD, n.OPT_PBWD_COOP):
            res = []
            for tail in (n.OPT_ZERO_RECORDS, n.OPT_TILE_CUTS):
                run.debug = walk | gather | tail
                res.append(run.backward(gC, gO))
            for k in res[0]:
                assert np.isfinite(res[1][k]).all(), k
                assert np.array_equal(res[0][k], res[1][k]), "%s: dL/d%s differs between zero records and tile cuts (walk %x, gather %x)" % (kind, k, walk, gather)


@pytest.mark.parametrize("shape,degree", [((1500, 208, 136), 3), ((4099, 320, 240), 3), ((64, 96, 80), 3), ((1500, 208, 136), 1), ((1500, 208, 136), 0)])
def test_preprocess_bwd_sh_direction_rows(shape, degree):
    """The view direction's share of dL/dmeans3D: preprocess_fwd leaves d(SH colour) / d(direction) (48 B per surfel) and preprocess_bwd
    contracts it with the colour gradient, instead of reading the 192-B SH block a second time.  Against the re-read form
    (SURFEL_OPT_PBWD_NO_JAC; a frame rendered with SURFEL_OPT_NO_STREAM has no rows and takes it by itself): every other gradient
    BIT-IDENTICAL, dL/dmeans3D equal to fp32 rounding; ragged last wave, culled surfels, lower SH degrees, zero records and tile cuts."""
    import surfel_native as n
    import synthetic
    P, W, H = shape
    sc = synthetic.make_scene(P, W, H, seed=P, px_radius=4.0)
    a = scene_args(sc)
    a["sh_degree"] = degree
    rng = np.random.default_rng(5)
    gC = rng.normal(size=(3, H, W)).astype(np.float32); gO = rng.normal(size=(7, H, W)).astype(np.float32)
    run = HipRun(a).forward()
    for tail in (n.OPT_ZERO_RECORDS, n.OPT_TILE_CUTS):
        res = []
        for jac in (n.OPT_PBWD_NO_JAC, 0):
            run.debug = n.OPT_PBWD_THREAD | tail | jac
            res.append(run.backward(gC, gO))
        for k in res[0]:
            assert np.isfinite(res[1][k]).all(), k
            if k == "means3D":
                ref = res[0][k].astype(np.float64); x = res[1][k].astype(np.float64)
                assert np.abs(x - ref).max() <= 2e-5 * np.abs(ref).max() + 1e-12, "dL/dmeans3D: rows vs re-read differ by %.3e" % np.abs(x - ref).max()
                assert cosine(x, ref) >= 1.0 - 1e-10
            else:
                assert np.array_equal(res[0][k], res[1][k]), "dL/d%s differs (tail %x)" % (k, tail)
    assert np.abs(res[0]["means3D"]).max() > 0
    # a frame whose forward left no rows (no-grad render) still gets the same backward as the re-read form
    run2 = HipRun(a, debug=n.OPT_NO_STREAM).forward()
    run2.debug = n.OPT_PBWD_THREAD | n.OPT_ZERO_RECORDS
    g2 = run2.backward(gC, gO)
    run.debug = n.OPT_PBWD_THREAD | n.OPT_ZERO_RECORDS | n.OPT_PBWD_NO_JAC
    g1 = run.backward(gC, gO)
    for k in g1:
        assert np.array_equal(g1[k], g2[k]), k


def test_heavy_surfels_are_gathered_by_the_wave():
    """Surfels with more than 128 instance records (background-sized discs covering hundreds of tiles) have their records summed by
    the whole wave in preprocess_bwd (lane-strided partial sums + butterfly) instead of by one thread walking them: gradients against
    the fp64 oracle, the two record gathers and the two tail forms (zero records / tile cuts) bit-identical to each other, and
    reproducible."""
    import surfel_native as n
    import synthetic
    from oracle.surfel_oracle import Oracle
    sc = synthetic.make_scene(1200, 512, 384, seed=13, px_radius=6.0, z_near=1.0, z_far=8.0)
    rng = np.random.default_rng(13)
    P = sc["means3D"].shape[0]
    big = rng.random(P) < 0.03                                   # a few dozen discs of 100-300 px radius, faint enough to see through
    sc["scales"][big] *= rng.choice([20.0, 40.0], size=(int(big.sum()), 1)).astype(np.float32)
    sc["opacities"][big] = 0.05
    a = scene_args(sc)
    gC = rng.normal(size=(3, a["H"], a["W"])).astype(np.float32); gO = rng.normal(size=(7, a["H"], a["W"])).astype(np.float32)
    run = HipRun(a).forward()
    touched = run.ga.last()                                       # white box: tiles_touched sits behind rec | depths in the geometry buffer
    off = ((P * 112 + 255) // 256 * 256) + ((4 * P + 255) // 256 * 256)
    tt = touched[off:off + 4 * P].view(run.torch.int32).cpu().numpy()
    assert (tt > 128).sum() >= 5, "scene has no heavy surfels (max %d instances)" % tt.max()
    res = []
    for flags in (n.OPT_PBWD_THREAD | n.OPT_ZERO_RECORDS, n.OPT_PBWD_COOP | n.OPT_ZERO_RECORDS, n.OPT_PBWD_THREAD | n.OPT_TILE_CUTS,
                  n.OPT_PBWD_COOP | n.OPT_TILE_CUTS, n.OPT_PBWD_THREAD | n.OPT_ZERO_RECORDS):
        run.debug = flags
        res.append(run.backward(gC, gO))
    for r in res[1:]:
        for k in res[0]:
            assert np.array_equal(res[0][k], r[k]), "dL/d%s differs between the gather / tail variants" % k
    # discs of hundreds of pixels are the ill-conditioned end of the algorithm in fp32: the module's bar on the fp32-determined elements
    det = D.Determinacy(a, run.depths(), n_draws=5)
    D.judge_images("heavy", det, run.color.cpu().numpy(), run.others.cpu().numpy())
    det.backward(gC, gO)
    D.judge_grads("heavy", det, res[0])


def _clustered_scene(P=30000, W=400, H=304, seed=17, shrink=0.3):
    """Surfels pulled towards the view axis: the image's centre tiles hold long lists, most tiles none (an object-centred frame)."""
    import synthetic
    sc = synthetic.make_scene(P, W, H, seed=seed, px_radius=6.0, z_near=2.0, z_far=8.0)
    V = sc["viewmatrix"].astype(np.float64)                      # row-vector convention: p_view = [p, 1] @ V
    pv = np.concatenate([sc["means3D"].astype(np.float64), np.ones((P, 1))], 1) @ V
    pv[:, :2] *= shrink
    sc["means3D"] = (pv @ np.linalg.inv(V))[:, :3].astype(np.float32)
    sc["opacities"] = np.full_like(sc["opacities"], 0.15)        # faint: long lists are walked to their end
    return sc


@pytest.mark.parametrize("kind", ["clustered", "uniform"])
def test_tile_order_is_scheduling_only(kind):
    """Which tile a blend workgroup takes (tile_order_kernel: XCD-contiguous runs | groups of 4 tiles, longest lists first, round-robin
    over the XCDs | decided per frame on the device) only schedules: images and gradients BIT-IDENTICAL under all three settings and
    every backward walk; the device picks the longest-first order for the clustered frame and the contiguous one for the uniform frame
    (white box: the map at the end of the image buffer)."""
    import surfel_native as n
    sc = _clustered_scene() if kind == "clustered" else _scene((30000, 400, 304), seed=17, px_radius=4.0)
    a = scene_args(sc)
    rng = np.random.default_rng(6)
    gC = rng.normal(size=(3, a["H"], a["W"])).astype(np.float32); gO = rng.normal(size=(7, a["H"], a["W"])).astype(np.float32)
    gx, gy = (a["W"] + 15) // 16, (a["H"] + 15) // 16
    map_len = 32 * ((((gx + 3) // 4) * gy + 7) // 8)
    res, maps = [], []
    for mode in (1, 2, 0):
        run = HipRun(a, debug=n.opt_tile_order(mode)).forward()
        buf = run.ia.last()
        maps.append(None)
        img = (run.color.cpu().numpy(), run.others.cpu().numpy(), run.radii.cpu().numpy(), run.R)
        gs = []
        for walk in (n.OPT_BWD_ROWS, n.OPT_BWD_QUAD, n.OPT_BWD_SCAN):
            run.debug = walk
            gs.append(run.backward(gC, gO))
        # white box (ImgState::carve): the flag = second word behind the 2 x 64 partial counters, the map = last allocation
        import diff_surfel_rasterization as dsr
        flag = int(buf[(gx * gy + 64) * 8 + 4:(gx * gy + 64) * 8 + 8].view(run.torch.int32).item())
        off = dsr.image_layout(a["W"], a["H"])[4]
        tm = buf[off:off + 4 * map_len].view(run.torch.int32).cpu().numpy()
        if flag:
            assert sorted(tm[tm >= 0].tolist()) == list(range(gx * gy)), "the map must hold every tile exactly once"
        maps[-1] = flag
        res.append((img, gs))
    (i0, g0) = res[0]
    for (i1, g1) in res[1:]:
        assert i0[3] == i1[3] and all(np.array_equal(x, y) for x, y in zip(i0[:3], i1[:3])), "images differ with the tile order"
        for ga, gb in zip(g0, g1):
            for k in ga:
                assert np.array_equal(ga[k], gb[k]), "dL/d%s differs with the tile order" % k
    contiguous, balanced, auto = maps
    assert contiguous == 0 and balanced == 1
    assert auto == (1 if kind == "clustered" else 0), "device-side choice for the %s frame" % kind


def test_capacity_binning_is_identical():
    """Capacity binning (binning buffers sized from the previous frames' instance counts, fused scan + emission + histogram kernel,
    sort passes / tile ranges with the count on the device, no host wait in the middle of the forward) against the exact-size path:
    instance count, radii, images and gradients BIT-IDENTICAL — on the steady-state frame, and on a frame that overflows its
    capacity and is redone (frame sizes no other test uses, so the history of this thread's capacity table is this test's own)."""
    import surfel_native as n
    import synthetic
    lib = n.load()

    def run(sc, flags):
        a = scene_args(sc)
        r = HipRun(a, debug=flags | n.opt_tile_sort(2)).forward()      # (per-tile depth sort forced: the capacity path's precondition)
        kind = lib.surfel_debug_last_binning()
        gC = np.random.default_rng(1).normal(size=(3, a["H"], a["W"])).astype(np.float32)
        gO = np.random.default_rng(2).normal(size=(7, a["H"], a["W"])).astype(np.float32)
        g = r.backward(gC, gO)
        return kind, (r.R, r.color.cpu().numpy(), r.others.cpu().numpy(), r.radii.cpu().numpy(), g)

    def same(x, y, what):
        assert x[0] == y[0], (what, x[0], y[0])
        assert np.array_equal(x[3], y[3]), what
        assert np.array_equal(x[1], y[1]) and np.array_equal(x[2], y[2]), "%s: images differ" % what
        for k in x[4]:
            assert np.array_equal(x[4][k], y[4][k]), "%s: dL/d%s differs" % (what, k)

    W, H = 304, 176
    sc = synthetic.make_scene(20000, W, H, seed=3, px_radius=4.0)
    k0, exact = run(sc, n.OPT_EXACT_BINNING)
    assert k0 == 0
    k1, first = run(sc, 0)            # the exact frame above left a count behind: capacity path at once
    k2, steady = run(sc, 0)
    assert k1 == 1 and k2 == 1, (k1, k2)
    same(first, exact, "first capacity frame"); same(steady, exact, "steady capacity frame")
    # overflow: four times the surfels at the same frame size
    sc2 = synthetic.make_scene(80000, W, H, seed=5, px_radius=4.0)
    k3, over = run(sc2, 0)
    assert k3 == 2, k3
    k4, exact2 = run(sc2, n.OPT_EXACT_BINNING)
    assert k4 == 0 and over[0] > 1.5 * exact[0]
    same(over, exact2, "overflowing frame")
    k5, after = run(sc2, 0)           # the capacity has grown
    assert k5 == 1
    same(after, exact2, "frame after the overflow")
    # an empty view (every surfel behind the camera) on the capacity path
    sc3 = dict(sc); sc3["means3D"] = (sc["means3D"] - 1e3 * (sc["means3D"] - sc["campos"][None])).astype(np.float32)
    k6, empty = run(sc3, 0)
    k7, empty_exact = run(sc3, n.OPT_EXACT_BINNING)
    assert k6 == 1 and empty[0] == 0
    same(empty, empty_exact, "empty frame")


def test_lazily_counted_forward():
    """SURFEL_OPT_LAZY_COUNT (include/surfel_hip.h): a capacity-path forward that returns its capacity without waiting for the instance
    count; surfel_forward_count() delivers the exact count later.  Images and — with the capacity passed on as num_rendered — the
    gradients are BIT-IDENTICAL to the waiting forward's; a frame that overflows its capacity is reported (SURFEL_E_OVERFLOW) and,
    if nobody asks, stops the next forward."""
    import surfel_native as n
    import synthetic
    lib = n.load()
    W, H = 336, 176      # (a frame size no other test uses: the capacity history is this test's own)

    def run(sc, flags):
        a = scene_args(sc)
        r = HipRun(a, debug=flags | n.opt_tile_sort(2)).forward()
        kind = lib.surfel_debug_last_binning()
        return a, r, kind

    def grads(a, r):
        gC = np.random.default_rng(1).normal(size=(3, a["H"], a["W"])).astype(np.float32)
        gO = np.random.default_rng(2).normal(size=(7, a["H"], a["W"])).astype(np.float32)
        return r.backward(gC, gO)

    sc = synthetic.make_scene(20000, W, H, seed=3, px_radius=4.0)
    a, exact, k0 = run(sc, n.OPT_EXACT_BINNING)
    assert k0 == 0 and n.forward_count() == exact.R
    g_exact = grads(a, exact)
    _, lazy0, k1 = run(sc, n.OPT_LAZY_COUNT)
    assert k1 == 4 and lazy0.R >= exact.R and lazy0.R % 16384 == 0, (k1, lazy0.R, exact.R)      # the capacity, not the count
    assert n.forward_count() == exact.R
    assert n.forward_count() == exact.R                       # (no pending frame: the last count again)
    assert np.array_equal(lazy0.color.cpu().numpy(), exact.color.cpu().numpy()) and np.array_equal(lazy0.others.cpu().numpy(), exact.others.cpu().numpy())
    g_lazy = grads(a, lazy0)
    for k in g_exact:
        assert np.array_equal(g_exact[k], g_lazy[k]), "dL/d%s differs with num_rendered = capacity" % k
    # a frame without the count collected: the next forward collects it silently
    _, lazy1, k2 = run(sc, n.OPT_LAZY_COUNT)
    _, lazy2, k3 = run(sc, n.OPT_LAZY_COUNT)
    assert (k2, k3) == (4, 4) and n.forward_count() == exact.R
    # overflow: four times the surfels at the same frame size
    sc2 = synthetic.make_scene(80000, W, H, seed=5, px_radius=4.0)
    _, over, k4 = run(sc2, n.OPT_LAZY_COUNT)
    assert k4 == 4
    # the backward of the overflowed frame BEFORE anybody collected its count (what Trainer.step does): num_rendered is the capacity,
    # the records' first-instance slots run past the gradient records sized from it — every backward kernel must return at once
    # (rows, quad, scan walks; both record gathers): the poisoned outputs stay untouched, nothing faults
    for bits in (n.OPT_BWD_ROWS, n.OPT_BWD_QUAD, n.OPT_BWD_SCAN, n.OPT_PBWD_COOP, n.OPT_TILE_CUTS):
        over.debug = n.OPT_LAZY_COUNT | n.opt_tile_sort(2) | bits
        g_over = grads(scene_args(sc2), over)
        for k, v in g_over.items():
            assert np.isnan(v).all(), "backward of an overflowed frame wrote dL/d%s (%s)" % (k, bits)
    with pytest.raises(n.CapacityOverflow):
        n.forward_count()
    a2, redo, k5 = run(sc2, n.OPT_EXACT_BINNING)
    assert k5 == 0 and redo.R > over.R
    _, lazy3, k6 = run(sc2, n.OPT_LAZY_COUNT)                 # the capacity has grown
    assert k6 == 4 and n.forward_count() == redo.R
    assert np.array_equal(lazy3.color.cpu().numpy(), redo.color.cpu().numpy())
    # an overflow nobody looked at stops the next forward
    sc3 = synthetic.make_scene(200000, W, H, seed=6, px_radius=4.0)
    _, over2, k7 = run(sc3, n.OPT_LAZY_COUNT)
    assert k7 == 4
    a3 = scene_args(sc3)
    with pytest.raises(AssertionError, match="overflowed"):
        HipRun(a3, debug=n.opt_tile_sort(2)).forward()
    _, ok, k8 = run(sc3, 0)                                   # reported once: the thread goes on
    assert k8 in (0, 1, 2) and ok.R > redo.R


@pytest.mark.parametrize("kind", ["plain", "needles", "huge", "tiny", "grazing", "opaque_faint", "huge_faint", "crowded", "saturating", "C1", "clustered", "long_lists"])
def test_forward_kernels_are_identical(kind):
    """blend_fwd with software-pipelined staging (batches of 128 instances, LDS-DMA of the next batch's records under the walk, one
    barrier per batch: blend_fwd_pipe_kernel, the default) against the batch-synchronous kernel it replaces ("fwd_pipe" = 0): same walk,
    same per-pair arithmetic -> all ten channels, the per-pixel contributor counts the backward starts from (white box) and therefore
    the gradients are BIT-IDENTICAL; lists of every length class (empty tiles, < 64, not a multiple of 128, several batches, tiles
    that saturate in the middle of a batch)."""
    import surfel_native as n
    import synthetic
    lib = n.load()
    if kind == "clustered":
        sc = _clustered_scene()
    elif kind == "long_lists":      # ~700 instances per tile: six batches of 128, the last one partial
        sc = synthetic.make_scene(60000, 160, 128, seed=12, px_radius=6.0, z_near=1.0, z_far=9.0)
        sc["opacities"] = np.full_like(sc["opacities"], 0.015)
    else:
        sc = _walk_scene(kind)
    a = scene_args(sc)
    rng = np.random.default_rng(9)
    gC = rng.normal(size=(3, a["H"], a["W"])).astype(np.float32); gO = rng.normal(size=(7, a["H"], a["W"])).astype(np.float32)
    out = []
    try:
        for pipe in (1, 0, 1):
            assert lib.surfel_set_option(b"fwd_pipe", pipe) == 0
            run = HipRun(a, debug=n.OPT_BWD_ROWS).forward()
            gx, gy = (a["W"] + 15) // 16, (a["H"] + 15) // 16
            import diff_surfel_rasterization as dsr
            off = dsr.image_layout(a["W"], a["H"])[2]
            state = run.ia.last()[off:off + 20 * a["W"] * a["H"]].clone().cpu().numpy()      # final_T, M1, M2 | last, median contributor
            out.append((run.R, run.color.cpu().numpy(), run.others.cpu().numpy(), state, run.backward(gC, gO)))
    finally:
        lib.surfel_set_option(b"fwd_pipe", 1)
    assert np.isfinite(out[0][1]).all() and np.isfinite(out[0][2]).all()
    for other in out[1:]:
        assert out[0][0] == other[0]
        for k in (1, 2, 3):
            assert np.array_equal(out[0][k], other[k]), "%s: forward output %d differs between the kernels" % (kind, k)
        for k in out[0][4]:
            assert np.array_equal(out[0][4][k], other[4][k]), "%s: dL/d%s differs" % (kind, k)


@pytest.mark.parametrize("kind", ["plain", "needles", "huge", "grazing", "huge_faint", "crowded", "saturating", "C1", "clustered", "long_lists"])
def test_tile_stream_is_staging_only(kind):
    """The TILE STREAM (csrc/surfel_common.h; round 5): blend_fwd leaves, per list position it walked, the 80-B blend record and the 16
    footprint bits in list order; blend_bwd (rows and scan walks) stages a batch from that contiguous stream instead of ids -> 112-B
    gather -> footprint test.  It only changes how a batch reaches LDS: for every walk the gradients with the stream are BIT-IDENTICAL
    to the gradients of the gathering staging (per call: SURFEL_OPT_BWD_GATHER; process-wide: "tile_stream" = 0 — the forward then
    writes none and the backward finds the two offset words zero), on the exact and on the capacity binning path, and for a forward
    by the batch-synchronous kernel (which writes no stream)."""
    import surfel_native as n
    import synthetic
    lib = n.load()
    if kind == "clustered":
        sc = _clustered_scene()
    elif kind == "long_lists":      # ~700 instances per tile: six batches of 128, the last one partial
        sc = synthetic.make_scene(60000, 160, 128, seed=12, px_radius=6.0, z_near=1.0, z_far=9.0)
        sc["opacities"] = np.full_like(sc["opacities"], 0.015)
    else:
        sc = _walk_scene(kind)
    a = scene_args(sc)
    rng = np.random.default_rng(19)
    gC = rng.normal(size=(3, a["H"], a["W"])).astype(np.float32); gO = rng.normal(size=(7, a["H"], a["W"])).astype(np.float32)
    walks = (("rows", n.OPT_BWD_ROWS), ("scan", n.OPT_BWD_SCAN), ("auto", 0))
    res = {}
    try:
        for tag, stream, pipe in (("stream", 1, 1), ("stream_again", 1, 1), ("off", 0, 1), ("old_forward", 1, 0)):
            assert lib.surfel_set_option(b"tile_stream", stream) == 0 and lib.surfel_set_option(b"fwd_pipe", pipe) == 0
            run = HipRun(a, debug=n.OPT_EXACT_BINNING if tag == "stream" else 0).forward()      # (the later forwards of this size take the capacity path)
            res[tag + "_binning"] = lib.surfel_debug_last_binning()
            for w, flag in walks:
                run.debug = flag
                res[(tag, w)] = run.backward(gC, gO)
                run.debug = 0
                if tag == "stream":
                    run.debug = flag | n.OPT_BWD_GATHER
                    res[("gather", w)] = run.backward(gC, gO)
    finally:
        lib.surfel_set_option(b"tile_stream", 1)
        lib.surfel_set_option(b"fwd_pipe", 1)
    assert res["stream_binning"] == 0      # (the later forwards take the capacity path where the frame qualifies for it: crowded tiles do not)
    assert res["stream_again_binning"] in (1, 2) or kind in ("huge_faint", "long_lists", "crowded", "saturating"), res["stream_again_binning"]
    for w, _ in walks:
        ref = res[("stream", w)]
        for k in ref:
            assert np.isfinite(ref[k]).all(), (w, k)
        for tag in ("gather", "stream_again", "off", "old_forward"):
            for k in ref:
                assert np.array_equal(ref[k], res[(tag, w)][k]), "%s, %s walk: dL/d%s with the tile stream differs from '%s'" % (kind, w, k, tag)


def _walk_scene(kind, seed=31):
    import synthetic
    if kind == "C1":
        return _scene("C1", seed=2)
    if kind == "crowded":
        sc = synthetic.make_scene(9000, 96, 64, seed=6, px_radius=30.0, z_near=2.0, z_far=8.0)
        sc["opacities"] = np.full_like(sc["opacities"], 0.02)
        return sc
    if kind == "saturating":      # opaque, deep: most lists are cut short by saturation (tile cuts, dead sub-tiles)
        sc = synthetic.make_scene(20000, 128, 96, seed=9, px_radius=9.0, z_near=1.0, z_far=9.0)
        sc["opacities"] = np.full_like(sc["opacities"], 0.95)
        return sc
    return _stress_scene(kind, seed)


@pytest.mark.parametrize("kind", ["plain", "needles", "huge", "tiny", "grazing", "opaque_faint", "crowded", "saturating", "C1"])
def test_scan_walk_matches_the_oracle_and_the_other_walks(kind):
    """blend_bwd scan walk (lanes = instances, DPP row scans for T and the suffix sum, gradients accumulated in registers;
    surfel_backward_scan.hip).  Its summation order differs from the rows / quad walks, so it cannot share their bits; it must
    (a) be bit-reproducible run to run, (b) meet the oracle bars of the element test wherever the rows walk meets them,
    (c) agree with the rows walk to fp32 summation noise: cosine >= 0.999999 and >= 99.9 % of elements within
    1e-5 mean|ref| + 1e-3 |ref|."""
    import surfel_native as n
    from oracle.surfel_oracle import Oracle
    a = scene_args(_walk_scene(kind))
    rng = np.random.default_rng(8)
    gC = rng.normal(size=(3, a["H"], a["W"])).astype(np.float32); gO = rng.normal(size=(7, a["H"], a["W"])).astype(np.float32)
    run = HipRun(a).forward()
    res = {}
    for name, flag in (("rows", n.OPT_BWD_ROWS), ("scan", n.OPT_BWD_SCAN), ("scan2", n.OPT_BWD_SCAN)):
        run.debug = flag
        res[name] = run.backward(gC, gO)
    o = Oracle("f64")
    R, col, oth, radii, st = oracle_forward(o, a, depth_key=run.depths())
    og = o.rasterize_backward(st, gC, gO)
    refs = dict(means3D=og.dL_dmeans3D, opacity=og.dL_dopacity, sh=og.dL_dsh, means2D=og.dL_dmean2D, scales=og.dL_dscales, rots=og.dL_drots)
    for k in res["rows"]:
        x, y = res["scan"][k], res["rows"][k]
        assert np.isfinite(x).all(), k
        assert np.array_equal(x, res["scan2"][k]), "scan walk not reproducible: %s" % k
        scale = np.abs(y).mean() + 1e-30
        f = frac_close(x, y, 1e-5 * scale, 1e-3)
        cs = cosine(x, y)
        assert f >= 0.999 and cs >= 0.999999, "%s: dL/d%s scan vs rows: %.5f of elements, cosine %.8f" % (kind, k, f, cs)
        if k in refs:
            ref = refs[k]
            sc_ = np.abs(ref).mean() + 1e-30
            fs = frac_close(x.reshape(ref.shape), ref, 1e-4 * sc_ + 1e-12, G_RTOL)
            fr = frac_close(y.reshape(ref.shape), ref, 1e-4 * sc_ + 1e-12, G_RTOL)
            assert fs >= fr - 5e-4, "%s: dL/d%s vs oracle: scan %.5f, rows %.5f" % (kind, k, fs, fr)


@pytest.mark.parametrize("kind", ["long_lists", "crowded", "huge_faint", "saturating", "C1"])
@pytest.mark.parametrize("gscale", [1000.0, 30000.0])
def test_distortion_dominated_gradients(kind, gscale):
    """The upstream-gradient RATIO of the reference's DTU configuration: with lambda_dist = 1000
    (/root/reference/scripts/dtu_eval.py:23, loss at /root/reference/train.py:77-88) dL/d(distortion map) is thousands of times the
    colour gradient, and the distortion term of a pair's u — mm^2 A - 2 mm M1 + M2, a variance-like form whose large terms cancel —
    dominates dL/dalpha.  Every walk against the fp64 oracle with N(0,1) gradients on nine channels and N(0,1) x gscale on the
    distortion channel, under the module's bar on the fp32-determined elements (determinacy.py)."""
    import surfel_native as n
    import synthetic
    from oracle.surfel_oracle import Oracle
    if kind == "long_lists":
        sc = synthetic.make_scene(60000, 160, 128, seed=12, px_radius=6.0, z_near=1.0, z_far=9.0)
        sc["opacities"] = np.full_like(sc["opacities"], 0.015)
    else:
        sc = _walk_scene(kind)
    a = scene_args(sc)
    rng = np.random.default_rng(8)
    gC = rng.normal(size=(3, a["H"], a["W"])).astype(np.float32); gO = rng.normal(size=(7, a["H"], a["W"])).astype(np.float32)
    gO[6] *= gscale
    run = HipRun(a).forward()
    det = D.Determinacy(a, run.depths(), n_draws=5)
    det.backward(gC, gO)
    names = tuple(x for x in D.GRADS if x[0] != "means2D")
    m = det.grad_masks(names)
    for walk, flag in (("rows", n.OPT_BWD_ROWS), ("quad", n.OPT_BWD_QUAD), ("scan", n.OPT_BWD_SCAN)):
        run.debug = flag
        g = run.backward(gC, gO)
        old_cap, D.EXEMPT_CAP = D.EXEMPT_CAP, STRESS_EXEMPT_CAP      # (built to be ill-conditioned: the probe may exempt more here)
        try:
            for k, _ in names:
                D.judge("g_dist x%g %s %s" % (gscale, kind, walk), k, g[k], *m[k], D.grad_tol(m[k][0]))
        finally:
            D.EXEMPT_CAP = old_cap


@pytest.mark.parametrize("kind", ["plain", "needles", "huge", "tiny", "opaque_faint", "huge_faint"])
def test_scan_walk_culling_is_exact(kind):
    """The scan walk with culling on / off: the cull only removes (pixel, surfel) pairs that contribute nothing, but it changes
    which instances share a chunk, i.e. the association of the T product and of the suffix sums — so the comparison is to
    summation noise, not bits: cosine >= 0.999999, >= 99.9 % of elements within 1e-5 mean + 1e-3 |ref|."""
    import surfel_native as n
    a = scene_args(_stress_scene(kind, 11))
    rng = np.random.default_rng(11)
    gC = rng.normal(size=(3, a["H"], a["W"])).astype(np.float32); gO = rng.normal(size=(7, a["H"], a["W"])).astype(np.float32)
    res = []
    for cull in (1, 0):
        run = HipRun(a, debug=0 if cull else n.OPT_NO_CULL).forward()
        run.debug = n.OPT_BWD_SCAN | (0 if cull else n.OPT_NO_CULL)
        res.append(run.backward(gC, gO))
    for k in res[0]:
        x, y = res[0][k], res[1][k]
        assert np.isfinite(x).all() and np.isfinite(y).all(), k
        scale = np.abs(y).mean() + 1e-30
        assert frac_close(x, y, 1e-5 * scale, 1e-3) >= 0.999 and cosine(x, y) >= 0.999999, (kind, k)


@pytest.mark.parametrize("kind", ["plain", "needles", "huge", "tiny", "grazing", "opaque_faint", "crowded", "saturating", "C1"])
def test_forward_and_backward_composite_the_same_pairs(kind):
    """The backward re-decides, pair by pair, what the forward composited (alpha >= 1/255, depth >= near, p2 != 0, position <= last).
    Both sides evaluate ONE definition of the intersection (surfel_common.h: pair_hit, contraction off) — so the number of pairs the
    instrumented forward composited must equal the number of lanes every backward walk treats as composited."""
    import torch
    import surfel_native as n
    lib = n.load()
    a = scene_args(_walk_scene(kind))
    rng = np.random.default_rng(1)
    gC = rng.normal(size=(3, a["H"], a["W"])).astype(np.float32); gO = rng.normal(size=(7, a["H"], a["W"])).astype(np.float32)
    st = torch.zeros(8, dtype=torch.int64, device="cuda:0")
    try:
        assert lib.surfel_debug_set_blend_stats(n.ptr(st)) == 0
        # (exact binning: on the capacity path a frame that overflows the capacity an earlier test left behind for this frame size is
        # blended twice — the truncated attempt would be counted as well)
        run = HipRun(a, debug=n.OPT_EXACT_BINNING).forward()
        fwd_pairs = int(st.cpu().numpy()[6])
        assert fwd_pairs > 0
        for name, flag in (("rows", n.OPT_BWD_ROWS), ("quad", n.OPT_BWD_QUAD), ("scan", n.OPT_BWD_SCAN)):
            st.zero_()
            run.debug = flag
            run.backward(gC, gO)
            got = int(st.cpu().numpy()[1])
            assert got == fwd_pairs, "%s / %s: forward composited %d pairs, the backward %d" % (kind, name, fwd_pairs, got)
    finally:
        lib.surfel_debug_set_blend_stats(None)


@pytest.mark.parametrize("kind", ["plain", "huge", "crowded", "C1"])
def test_record_gather_variants_are_identical(kind):
    """preprocess_bwd sums a surfel's instance gradient records either per thread or wave-cooperatively (groups of 5 lanes read
    one 80-B record; chosen by R / P): both add every value over the records in emission order -> BIT-IDENTICAL gradients."""
    import surfel_native as n
    import synthetic
    if kind == "C1":
        sc = _scene("C1", seed=4)
    elif kind == "crowded":
        sc = synthetic.make_scene(9000, 96, 64, seed=6, px_radius=30.0, z_near=2.0, z_far=8.0)
        sc["opacities"] = np.full_like(sc["opacities"], 0.02)
    else:
        sc = _stress_scene(kind, 41)
    a = scene_args(sc)
    rng = np.random.default_rng(2)
    gC = rng.normal(size=(3, a["H"], a["W"])).astype(np.float32); gO = rng.normal(size=(7, a["H"], a["W"])).astype(np.float32)
    run = HipRun(a).forward()
    res = []
    for flag in (n.OPT_PBWD_THREAD, n.OPT_PBWD_COOP):
        run.debug = flag
        res.append(run.backward(gC, gO))
    for k in res[0]:
        assert np.isfinite(res[1][k]).all(), k
        assert np.array_equal(res[0][k], res[1][k]), "%s: dL/d%s differs between the record-gather variants" % (kind, k)


def test_optional_gradient_outputs_may_be_null():
    """dL_dnormal and (without transMat_precomp) dL_dtransMat are intermediates no caller of the Python API receives: NULL skips the
    stores and leaves every other output's bits alone; with transMat_precomp a NULL dL_dtransMat is an error."""
    import surfel_native as n
    sc = _stress_scene("plain", 35)
    a = scene_args(sc)
    rng = np.random.default_rng(5)
    gC = rng.normal(size=(3, a["H"], a["W"])).astype(np.float32); gO = rng.normal(size=(7, a["H"], a["W"])).astype(np.float32)
    run = HipRun(a).forward()
    full = run.backward(gC, gO)
    part = run.backward(gC, gO, skip=("normal", "transMat"))
    assert set(full) - set(part) == {"normal", "transMat"}
    for k in part:
        assert np.array_equal(full[k], part[k], equal_nan=True), k
    assert np.isfinite(full["transMat"]).all() and np.abs(full["normal"]).max() > 0


def test_backward_hook_splits_off_the_colour_gradients():
    """surfel_set_backward_hook: with a hook installed the backward finalises dL/dcolour in a kernel of its own, calls the hook
    (where a multi-GPU caller starts its all-gather), then runs the chain rule without touching dL/dcolour again — every output
    keeps the bits of the single-kernel path, with SH gradients and with precomputed colours."""
    import surfel_native as n
    sc = _stress_scene("plain", 33)
    a = scene_args(sc)
    rng = np.random.default_rng(4)
    gC = rng.normal(size=(3, a["H"], a["W"])).astype(np.float32); gO = rng.normal(size=(7, a["H"], a["W"])).astype(np.float32)
    P = sc["means3D"].shape[0]
    for kw in ({}, {"colors_precomp": rng.uniform(0, 1, (P, 3)).astype(np.float32)}):
        run = HipRun(a, **kw).forward()
        ref = run.backward(gC, gO)
        calls = []
        # the hook snapshots dL/dcolour on the stream: it must already be final there (that is what goes on the wire)
        n.set_backward_hook(lambda: calls.append(run.g["colors"].clone()))
        try:
            got = run.backward(gC, gO)
        finally:
            n.set_backward_hook(None)
        assert len(calls) == 1 and np.array_equal(calls[0].cpu().numpy(), ref["colors"])
        calls = [1]
        for k in ref:
            assert np.array_equal(ref[k], got[k], equal_nan=True), k      # (outputs the call does not produce keep their NaN poison)
        assert np.abs(got["colors"]).max() > 0
        again = run.backward(gC, gO)          # hook removed: the single-kernel path again
        assert calls == [1] and all(np.array_equal(ref[k], again[k], equal_nan=True) for k in ref)


@pytest.mark.parametrize("kind,expect", [("small_footprints", "rows"), ("wide_footprints", "scan")])
def test_auto_walk_follows_the_device_rule(kind, expect):
    """bwd_variant = auto: the rows walk and the scan walk are both launched and the DEVICE decides from the frame's own totals which one
    runs (>= 6 tile instances per emitting surfel -> scan; csrc/surfel_blend_bwd.h: device_picks_scan) — no timed probes, no history:
    the gradients of every auto call are the bits of the walk the rule names, call after call, on the exact and the capacity path."""
    import surfel_native as n
    lib = n.load()
    sc = _scene((6000, 208, 144), seed=5) if kind == "small_footprints" else _scene((3000, 208, 144), seed=5, px_radius=14.0)
    a = scene_args(sc)
    rng = np.random.default_rng(2)
    gC = rng.normal(size=(3, a["H"], a["W"])).astype(np.float32); gO = rng.normal(size=(7, a["H"], a["W"])).astype(np.float32)
    run = HipRun(a).forward()
    vis = int((run.radii > 0).sum().item())
    print("%s: %d instances, %d emitting surfels (%.1f per surfel)" % (kind, run.R, vis, run.R / max(1, vis)))
    assert (run.R >= 6 * vis) == (expect == "scan")
    refs = {}
    for w, flag in (("rows", n.OPT_BWD_ROWS), ("scan", n.OPT_BWD_SCAN)):
        run.debug = flag
        refs[w] = run.backward(gC, gO)
    assert any(not np.array_equal(refs["rows"][k], refs["scan"][k]) for k in refs["rows"])      # (the two walks do differ in their bits)
    for it in range(4):
        run = HipRun(a).forward()      # (later forwards of this size take the capacity path)
        g = run.backward(gC, gO)
        for k in g:
            assert np.array_equal(refs[expect][k], g[k]), (it, k, lib.surfel_debug_last_binning())


def test_blend_stats_counters():
    """surfel_debug_set_blend_stats: the instrumented kernels count lane slots issued and lanes that held a composited pair;
    the per-row walk must waste fewer lanes than the per-quad walk on small footprints and see exactly the same useful pairs."""
    import torch
    import surfel_native as n
    lib = n.load()
    sc = _scene("C1", seed=3)
    a = scene_args(sc)
    rng = np.random.default_rng(1)
    gC = rng.normal(size=(3, a["H"], a["W"])).astype(np.float32); gO = rng.normal(size=(7, a["H"], a["W"])).astype(np.float32)
    run = HipRun(a).forward()
    out = {}
    try:
        for name, flag in (("rows", n.OPT_BWD_ROWS), ("quad", n.OPT_BWD_QUAD)):
            st = torch.zeros(8, dtype=torch.int64, device="cuda:0")
            assert lib.surfel_debug_set_blend_stats(n.ptr(st)) == 0
            run.debug = flag
            run.backward(gC, gO)
            out[name] = st.cpu().numpy().copy()
    finally:
        lib.surfel_debug_set_blend_stats(None)
    assert out["rows"][1] == out["quad"][1] > 0                 # identical set of composited pairs
    fr, fq = out["rows"][1] / out["rows"][0], out["quad"][1] / out["quad"][0]
    print("useful lanes: rows %.3f (%d wave visits), quad %.3f (%d wave visits)" % (fr, out["rows"][2], fq, out["quad"][2]))
    assert fr > fq and out["rows"][2] < out["quad"][2]


@pytest.mark.parametrize("n,bits", [(1, (0, 32)), (63, (0, 32)), (2047, (0, 12)), (2048, (0, 32)), (2049, (3, 17)), (300_000, (0, 32)),
                                    (611_573, (0, 12)), (3_000_001, (0, 32)), (5_000_000, (0, 15)), (2_000_003, (7, 8)), (2_000_003, (5, 5)),
                                    (4097, (31, 32)), (4097, (0, 0)), (8191, (0, 32)), (8193, (0, 12)), (40_000, (0, 12)), (1_048_576, (0, 13))])
def test_radix_sort_is_stable_and_exact(n, bits):
    """The one-launch-per-pass radix sort (decoupled look-back, 8192-item tiles staged through LDS) against
    numpy's stable argsort, incl. heavy duplicates; 1-bit and empty key fields (the order is then the input order); large inputs
    also through rocprim::radix_sort_pairs."""
    import torch
    import surfel_native as nat
    lib = nat.load()
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(n)
    lo, hi = bits
    # large inputs: the library's own passes and rocprim::radix_sort_pairs; small ones: the look-back passes and rocPRIM (3: rocPRIM at every size)
    impls = (0, 1) if n > (1 << 20) else (0, 3)
    for trial in [(t, i) for i in impls for t in range(3)]:
        trial, impl = trial
        assert lib.surfel_set_option(b"large_sort", impl) == 0
        if trial == 0:
            keys = rng.integers(0, 2 ** 32, n, dtype=np.uint64).astype(np.uint32)
        elif trial == 1:   # few distinct keys: long same-digit runs, every tie must keep input order
            keys = rng.choice(rng.integers(0, 2 ** 32, 7, dtype=np.uint64).astype(np.uint32), n)
        else:              # float depth bits, like the hot path
            keys = rng.uniform(0.2, 60.0, n).astype(np.float32).view(np.uint32)
        vals = np.arange(n, dtype=np.uint32)
        k = torch.from_numpy(keys.view(np.int32)).to(dev); v = torch.from_numpy(vals.view(np.int32)).to(dev)
        alloc = nat.TorchAllocator(dev)
        rc = lib.surfel_debug_sort_pairs(alloc.cb, None, nat.ptr(k), nat.ptr(v), n, lo, hi, nat.current_stream_ptr(dev))
        assert rc == 0, nat.last_error()
        torch.cuda.synchronize()
        field = (keys >> np.uint32(min(lo, 31))) & np.uint32((1 << (hi - lo)) - 1 if hi - lo < 32 else 0xffffffff)
        order = np.argsort(field, kind="stable")
        assert np.array_equal(v.cpu().numpy().view(np.uint32), vals[order]), "trial %d impl %d: order differs" % (trial, impl)
        assert np.array_equal(k.cpu().numpy().view(np.uint32), keys[order])
    lib.surfel_set_option(b"large_sort", LARGE_SORT_DEFAULT)


def test_grad_arena_is_zero_copy_and_identical():
    """multi-GPU path: gradients written straight into surfel_dist.GradBucket's flat buffer == the default tensors."""
    import torch
    import diff_surfel_rasterization as d
    import surfel_dist as sd
    sc = _scene((900, 112, 80), seed=9, px_radius=5.0)
    dev = torch.device("cuda:0")
    t = lambda x: torch.as_tensor(np.ascontiguousarray(x)).to(dev)
    rs = d.GaussianRasterizationSettings(image_height=sc["H"], image_width=sc["W"], tanfovx=sc["tanfovx"], tanfovy=sc["tanfovy"], bg=t(sc["bg"]),
                                         scale_modifier=1.0, viewmatrix=t(sc["viewmatrix"]), projmatrix=t(sc["projmatrix"]), sh_degree=3,
                                         campos=t(sc["campos"]), prefiltered=False, debug=False)
    P = sc["means3D"].shape[0]
    g = torch.Generator(device="cpu").manual_seed(3)
    gC = torch.randn((3, sc["H"], sc["W"]), generator=g).to(dev); gO = torch.randn((7, sc["H"], sc["W"]), generator=g).to(dev)

    def run():
        ps = [t(sc[k]).requires_grad_(True) for k in ("means3D", "shs", "opacities", "scales", "rotations")]
        m2 = torch.zeros_like(ps[0], requires_grad=True)
        col, radii, allmap = d.GaussianRasterizer(rs)(means3D=ps[0], means2D=m2, shs=ps[1], colors_precomp=None, opacities=ps[2],
                                                     scales=ps[3], rotations=ps[4], cov3D_precomp=None)
        torch.autograd.backward([col, allmap], [gC, gO])
        return [p.grad for p in ps]
    ref = [x.clone() for x in run()]
    bucket = sd.GradBucket(P, dev)
    bucket.buf.fill_(float("nan"))
    try:
        d.set_grad_arena(bucket.arena())
        got = run()
    finally:
        d.set_grad_arena(None)
    names = ["xyz", "sh", "opacity", "scaling", "rotation"]
    for x, r, nm in zip(got, ref, names):
        assert torch.equal(x, r), nm
        # the kernel wrote the bucket section itself (autograd may or may not hand the same storage on as .grad)
        assert torch.equal(bucket.views[nm].reshape(r.shape), r), "%s: bucket section differs" % nm
    assert torch.isfinite(bucket.buf).all()          # every bucket element was written by the kernels


def test_backward_is_bit_reproducible():
    sc = _scene("C1", seed=1)
    a = scene_args(sc)
    rng = np.random.default_rng(0)
    gC = rng.normal(size=(3, a["H"], a["W"])).astype(np.float32); gO = rng.normal(size=(7, a["H"], a["W"])).astype(np.float32)
    outs = []
    for _ in range(3):
        run = HipRun(a).forward()
        outs.append(run.backward(gC, gO))
    for k in outs[0]:
        assert np.array_equal(outs[0][k], outs[1][k]) and np.array_equal(outs[0][k], outs[2][k]), k


def test_edge_cases():
    """empty scene, everything culled, single surfel, non-multiple-of-16 image."""
    sc = _scene((16, 40, 24), seed=0)
    a = scene_args(sc)
    # all behind the camera
    a2 = dict(a); a2["means3D"] = a["means3D"] * np.array([1, 1, -1], np.float32) - np.array([0, 0, 5], np.float32)
    run = HipRun(a2).forward()
    assert run.R == 0 and int(run.radii.abs().sum()) == 0
    assert np.allclose(run.others.cpu().numpy(), 0.0)
    assert np.allclose(run.color.cpu().numpy(), np.broadcast_to(a["bg"][:, None, None], (3, a["H"], a["W"])))
    g = run.backward(np.ones((3, a["H"], a["W"]), np.float32), np.ones((7, a["H"], a["W"]), np.float32))
    assert all(np.all(v == 0) for v in g.values())
    # P = 0
    a3 = dict(a)
    for k in ("means3D", "opacities", "scales", "rotations", "shs"):
        a3[k] = a[k][:0]
    run = HipRun(a3).forward()
    assert run.R == 0
    # single surfel
    from oracle.surfel_oracle import Oracle
    a4 = dict(a)
    for k in ("means3D", "opacities", "scales", "rotations", "shs"):
        a4[k] = a[k][:1]
    run = HipRun(a4).forward()
    R, col, oth, radii, st = oracle_forward(Oracle("f64"), a4)
    assert run.R <= R
    assert np.allclose(run.color.cpu().numpy(), col, atol=1e-4)


def test_argument_errors():
    import surfel_native as n
    sc = _scene((16, 40, 24), seed=0)
    a = scene_args(sc)
    run = HipRun(a)
    run.colors = run.means3D          # both shs and colors_precomp -> must be rejected like the reference
    with pytest.raises(AssertionError, match="exactly one"):
        run.forward()
    run = HipRun(a); run.scales = None
    with pytest.raises(AssertionError, match="exactly one"):
        run.forward()
    assert n.load().surfel_abi_version() == 1


def test_dropin_autograd_module():
    """The reference-shaped Python surface: GaussianRasterizationSettings / GaussianRasterizer + autograd."""
    import torch
    from diff_surfel_rasterization import GaussianRasterizationSettings, GaussianRasterizer
    from oracle.surfel_oracle import Oracle
    sc = _scene((600, 80, 64), seed=6, px_radius=5.0)
    a = scene_args(sc)
    dev = torch.device("cuda:0")
    t = lambda x: torch.tensor(x, device=dev)
    rs = GaussianRasterizationSettings(image_height=a["H"], image_width=a["W"], tanfovx=a["tanfovx"], tanfovy=a["tanfovy"],
                                       bg=t(a["bg"]), scale_modifier=1.0, viewmatrix=t(a["viewmatrix"]),
                                       projmatrix=t(a["projmatrix"]), sh_degree=3, campos=t(a["campos"]), prefiltered=False,
                                       debug=False)
    rast = GaussianRasterizer(raster_settings=rs)
    means3D = t(a["means3D"]).requires_grad_(True); shs = t(a["shs"]).requires_grad_(True)
    opac = t(a["opacities"]).requires_grad_(True); scales = t(a["scales"]).requires_grad_(True)
    rots = t(a["rotations"]).requires_grad_(True)
    means2D = torch.zeros_like(means3D, requires_grad=True) + 0
    means2D.retain_grad()
    color, radii, allmap = rast(means3D=means3D, means2D=means2D, shs=shs, colors_precomp=None, opacities=opac, scales=scales,
                                rotations=rots, cov3D_precomp=None)
    assert color.shape == (3, a["H"], a["W"]) and allmap.shape == (7, a["H"], a["W"]) and radii.shape == (600,)
    rng = np.random.default_rng(3)
    gC = rng.normal(size=color.shape).astype(np.float32); gO = rng.normal(size=allmap.shape).astype(np.float32)
    ((color * t(gC)).sum() + (allmap * t(gO)).sum()).backward()
    o = Oracle("f64")
    R, col, oth, rad, st = oracle_forward(o, a)
    og = o.rasterize_backward(st, gC, gO)
    assert means2D.grad is not None and means2D.grad.shape == (600, 3)
    assert cosine(means3D.grad.cpu().numpy(), og.dL_dmeans3D) > 0.999
    assert cosine(means2D.grad.cpu().numpy(), og.dL_dmean2D) > 0.999
    assert cosine(shs.grad.cpu().numpy(), og.dL_dsh) > 0.999
    assert cosine(opac.grad.cpu().numpy(), og.dL_dopacity) > 0.999
    assert cosine(scales.grad.cpu().numpy(), og.dL_dscales) > 0.999
    assert cosine(rots.grad.cpu().numpy(), og.dL_drots) > 0.999
    vis = rast.markVisible(means3D.detach())
    assert np.array_equal(vis.cpu().numpy(), o.mark_visible(a["means3D"], a["viewmatrix"]))
    with pytest.raises(Exception):
        rast(means3D=means3D, means2D=means2D, shs=shs, colors_precomp=shs, opacities=opac, scales=scales, rotations=rots)


def test_knn_exact():
    import torch
    from simple_knn._C import distCUDA2
    from oracle.surfel_oracle import Oracle
    rng = np.random.default_rng(0)
    for P in (7, 1000, 5000):
        pts = rng.normal(size=(P, 3)).astype(np.float32)
        if P == 1000:
            pts[:10] = pts[10:20]            # exact duplicates -> zero distances
        got = distCUDA2(torch.tensor(pts, device="cuda:0")).cpu().numpy()
        ref = Oracle("f64").knn_dist2(pts)
        assert np.allclose(got, ref, rtol=1e-5, atol=1e-9), P
    # larger, clustered: check against scipy's kd-tree
    from scipy.spatial import cKDTree
    P = 200_000
    pts = (rng.normal(size=(P, 3)) * rng.choice([0.05, 1.0, 5.0], size=(P, 1))).astype(np.float32)
    got = distCUDA2(torch.tensor(pts, device="cuda:0")).cpu().numpy()
    d, _ = cKDTree(pts.astype(np.float64)).query(pts.astype(np.float64), k=4)
    ref = (d[:, 1:] ** 2).mean(1)
    assert np.allclose(got, ref, rtol=1e-4, atol=1e-9)


def test_tile_band_sharding_on_device():
    """surfel_dist.band_settings on the HIP path: bands concatenate to the full image and band gradients add up."""
    import torch
    import surfel_dist as sd
    from diff_surfel_rasterization import GaussianRasterizationSettings, GaussianRasterizer
    sc = _scene((4000, 160, 128), seed=8, px_radius=5.0)
    a = scene_args(sc)
    dev = torch.device("cuda:0")
    t = lambda x: torch.tensor(x, device=dev)
    rs = GaussianRasterizationSettings(image_height=a["H"], image_width=a["W"], tanfovx=a["tanfovx"], tanfovy=a["tanfovy"],
                                       bg=t(a["bg"]), scale_modifier=1.0, viewmatrix=t(a["viewmatrix"]),
                                       projmatrix=t(a["projmatrix"]), sh_degree=3, campos=t(a["campos"]), prefiltered=False, debug=False)
    rng = np.random.default_rng(4)
    gC = t(rng.normal(size=(3, a["H"], a["W"])).astype(np.float32)); gO = t(rng.normal(size=(7, a["H"], a["W"])).astype(np.float32))
    gO[5] = 0

    def run(settings, gc, go):
        leaves = [t(a[k]).requires_grad_(True) for k in ("means3D", "shs", "opacities", "scales", "rotations")]
        m2 = torch.zeros_like(leaves[0], requires_grad=True)
        color, radii, allmap = GaussianRasterizer(settings)(means3D=leaves[0], means2D=m2, shs=leaves[1], colors_precomp=None,
                                                           opacities=leaves[2], scales=leaves[3], rotations=leaves[4], cov3D_precomp=None)
        torch.autograd.backward([color, allmap], [gc, go])
        return color.detach(), allmap.detach(), [l.grad for l in leaves]

    col, oth, g_full = run(rs, gC, gO)
    acc = None
    for (y0, y1) in sd.band_bounds(a["H"], 4):
        cb, ob, gb = run(sd.band_settings(rs, y0, y1), gC[:, y0:y1].contiguous(), gO[:, y0:y1].contiguous())
        assert frac_close(cb.cpu().numpy(), col[:, y0:y1].cpu().numpy(), 2e-4, 2e-4) > 0.999
        assert frac_close(ob[:5].cpu().numpy(), oth[:5, y0:y1].cpu().numpy(), 2e-4, 2e-4) > 0.999
        acc = gb if acc is None else [x + y for x, y in zip(acc, gb)]
    for got, ref in zip(acc, g_full):
        assert cosine(got.cpu().numpy(), ref.cpu().numpy()) > 0.9999


def test_interleaved_forwards_and_backwards():
    """SURVEY 8b's threading convention: forward -> backward state travels with the call's own buffers, so several forwards may sit
    between a forward and its backward (/root/reference/train.py:211: training_report re-renders under no_grad between iterations;
    any caller may hold two graphs).  The library's process-wide state that could break this — the tile-stream registry, the lazy-count
    word, the per-size capacity history, the backward-walk word in the image buffer — is exercised: forward A (800x800), forward B
    (another size, on a side stream), a no_grad re-render of A and of B (no stream left behind: SURFEL_OPT_NO_STREAM), backward A,
    backward B (side stream).  Images and every gradient must equal the un-interleaved run BIT FOR BIT, through the drop-in autograd
    module and through the C ABI."""
    import torch
    import synthetic
    from diff_surfel_rasterization import GaussianRasterizationSettings, GaussianRasterizer
    dev = torch.device("cuda:0")
    t = lambda x: torch.tensor(x, device=dev)
    scenes = {"A": scene_args(synthetic.make_scene(40_000, 800, 800, seed=21, px_radius=5.0)),
              "B": scene_args(synthetic.make_scene(9_000, 336, 208, seed=22, px_radius=6.0))}
    rngs = np.random.default_rng(4)
    ups = {k: (t(rngs.normal(size=(3, a["H"], a["W"])).astype(np.float32)), t(rngs.normal(size=(7, a["H"], a["W"])).astype(np.float32))) for k, a in scenes.items()}

    def module_call(a, grad=True):
        rs = GaussianRasterizationSettings(image_height=a["H"], image_width=a["W"], tanfovx=a["tanfovx"], tanfovy=a["tanfovy"], bg=t(a["bg"]),
                                           scale_modifier=1.0, viewmatrix=t(a["viewmatrix"]), projmatrix=t(a["projmatrix"]), sh_degree=3,
                                           campos=t(a["campos"]), prefiltered=False, debug=False)
        p = {k: t(a[k]).requires_grad_(grad) for k in ("means3D", "shs", "opacities", "scales", "rotations")}
        m2 = torch.zeros_like(p["means3D"], requires_grad=grad)
        color, radii, allmap = GaussianRasterizer(raster_settings=rs)(means3D=p["means3D"], means2D=m2, shs=p["shs"], colors_precomp=None, opacities=p["opacities"],
                                                                      scales=p["scales"], rotations=p["rotations"], cov3D_precomp=None)
        p["means2D"] = m2
        return p, color, radii, allmap

    def finish(p, color, allmap, up):
        torch.autograd.backward([color, allmap], [up[0], up[1]])
        torch.cuda.synchronize()
        return {k: v.grad.detach().cpu().numpy().copy() for k, v in p.items()}, color.detach().cpu().numpy().copy(), allmap.detach().cpu().numpy().copy()

    # un-interleaved
    base = {}
    for k, a in scenes.items():
        p, color, radii, allmap = module_call(a)
        base[k] = finish(p, color, allmap, ups[k])
    # interleaved
    side = torch.cuda.Stream(device=dev)
    pA, cA, rA, mA = module_call(scenes["A"])
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):
        pB, cB, rB, mB = module_call(scenes["B"])
    with torch.no_grad():      # train.py:211-style re-renders between a forward and its backward
        _, c2, _, m2_ = module_call(scenes["A"], grad=False)
        _, c3, _, m3_ = module_call(scenes["B"], grad=False)
    torch.cuda.synchronize()
    assert np.array_equal(c2.cpu().numpy(), base["A"][1]) and np.array_equal(m3_.cpu().numpy(), base["B"][2])
    gA = finish(pA, cA, mA, ups["A"])
    with torch.cuda.stream(side):
        gB = finish(pB, cB, mB, ups["B"])
    for k, got in (("A", gA), ("B", gB)):
        assert np.array_equal(got[1], base[k][1]) and np.array_equal(got[2], base[k][2]), "%s: images differ when interleaved" % k
        for name in got[0]:
            assert np.array_equal(got[0][name], base[k][0][name]), "%s: dL/d%s differs when interleaved" % (k, name)

    # the same through the C ABI (HipRun keeps each call's three buffers), backward order swapped as well
    gup = {k: (ups[k][0].cpu().numpy(), ups[k][1].cpu().numpy()) for k in ups}
    ref = {}
    for k, a in scenes.items():
        r = HipRun(a).forward()
        ref[k] = (r.color.cpu().numpy(), r.others.cpu().numpy(), r.backward(*gup[k]))
    rA_, rB_ = HipRun(scenes["A"]).forward(), HipRun(scenes["B"]).forward()
    rA2 = HipRun(scenes["A"], debug=n_opt("OPT_NO_STREAM")).forward()      # a render-only frame of A's size in between
    gB2 = rB_.backward(*gup["B"])
    gA2 = rA_.backward(*gup["A"])
    assert np.array_equal(rA2.color.cpu().numpy(), ref["A"][0])
    for k, r, g in (("A", rA_, gA2), ("B", rB_, gB2)):
        assert np.array_equal(r.color.cpu().numpy(), ref[k][0]) and np.array_equal(r.others.cpu().numpy(), ref[k][1])
        for name in g:
            assert np.array_equal(g[name], ref[k][2][name]), "C ABI %s: dL/d%s differs when interleaved" % (k, name)
    # a frame rendered WITHOUT a stream (SURFEL_OPT_NO_STREAM: "no backward will follow") still has a correct backward: it gathers by
    # surfel id — same bits — and re-derives the view direction's share of dL/dmeans3D from the SH block instead of the rows the forward
    # did not leave: that tensor equal to fp32 rounding
    gA3 = rA2.backward(*gup["A"])
    for name in gA3:
        if name == "means3D":
            d = np.abs(gA3[name].astype(np.float64) - ref["A"][2][name])
            assert d.max() <= 2e-5 * np.abs(ref["A"][2][name]).max(), "no-stream frame: dL/dmeans3D off by %.3e" % d.max()
        else:
            assert np.array_equal(gA3[name], ref["A"][2][name]), "no-stream frame: dL/d%s differs" % name


def n_opt(name):
    import surfel_native as n
    return getattr(n, name)
