"""-m gpu: the HIP rasterizer (through the C ABI) against the ANALYTIC anchors of tests/analytic.py — the independent ray/plane
renderer and the closed forms for fronto-parallel discs on the optical axis.  These do not go through oracle/ at all, so they
anchor the product directly to the paper's geometry (the oracle is held to the same anchors in tests/test_analytic_cpu.py).

fp32 tolerance: images |d| <= 1e-4 + 1e-4 |ref| on >= 99.9 % of pixels (a pixel whose pair sits on a threshold may flip);
closed-form gradients to 2e-3 relative.
"""
import numpy as np
import pytest

import analytic
from helpers import HipRun, frac_close, scene_args

pytestmark = pytest.mark.gpu


def _images_close(run, ec, eo, frac=0.999):
    c = run.color.cpu().numpy(); o = run.others.cpu().numpy()
    assert frac_close(c, ec, 1e-4, 1e-4) >= frac, "color"
    for ch, nm in enumerate(["depth-sum", "alpha", "nx", "ny", "nz", "median", "distortion"]):
        f = frac_close(o[ch], eo[ch], 1e-4, 1e-4)
        assert f >= frac, "%s: %.5f of pixels within tolerance" % (nm, f)


@pytest.mark.parametrize("seed,tilt,P", [(1, True, 14), (2, False, 9), (4, True, 30), (6, True, 40)])
def test_hip_matches_ray_plane_renderer(seed, tilt, P):
    import synthetic
    W, H = 72, 56
    sc = synthetic.make_scene(P, W, H, seed=seed, px_radius=7.0, z_near=2.0, z_far=7.0, tilt=tilt)
    rng = np.random.default_rng(seed)
    sc["bg"] = np.array([0.3, 0.1, 0.6], np.float32)
    sc["opacities"] = rng.uniform(0.05, 0.95, (P, 1)).astype(np.float32)
    cols = rng.uniform(0, 1, (P, 3)).astype(np.float32)
    run = HipRun(scene_args(sc), colors_precomp=cols).forward()
    rc, ro, rr, rxy = analytic.raycast_render(sc, cols)
    assert np.array_equal(run.radii.cpu().numpy(), rr)
    # view depths the device sorts on = (mean, 1) @ viewmatrix, z
    d = (np.c_[sc["means3D"].astype(np.float64), np.ones(P)] @ sc["viewmatrix"].astype(np.float64))[:, 2]
    vis = rr > 0
    assert np.allclose(run.depths()[vis], d[vis], rtol=3e-7, atol=0)
    _images_close(run, rc, ro)


def test_hip_disc_on_axis_known_answer():
    W, H = 64, 48
    sc = analytic.axis_scene(W, H, [(3.0, 0.12, 0.07, 0.8, (0.9, 0.4, 0.2))])
    run = HipRun(scene_args(sc)).forward()
    alpha, hit, use3d, r, _ = analytic.disc_alpha(sc, 0)
    assert int(run.radii[0]) == r
    ec, eo = analytic.stacked_discs(sc)
    _images_close(run, ec, eo, frac=1.0)
    o = run.others.cpu().numpy()
    assert np.all(o[5][hit] == np.float32(3.0)) and np.all(o[5][~hit] == 0)       # median depth = z exactly where the disc contributes
    assert np.all(o[2] == 0) and np.all(o[3] == 0) and np.all(o[6] == 0)            # view normal (0, 0, -alpha); one disc: no distortion
    rng = np.random.default_rng(0)
    gC = rng.normal(size=(3, H, W)).astype(np.float32)
    g = run.backward(gC, np.zeros((7, H, W), np.float32))
    e = analytic.single_disc_grads(sc, gC.astype(np.float64))
    rel = lambda x, y: abs(float(x) - y) / abs(y)
    assert rel(g["opacity"][0, 0], e["opacity"]) < 2e-3
    assert np.abs(g["sh"][0, 0] - e["sh_dc"]).max() / np.abs(e["sh_dc"]).max() < 2e-3
    assert rel(g["scales"][0, 0], e["scale_u"]) < 2e-3 and rel(g["scales"][0, 1], e["scale_v"]) < 2e-3
    assert rel(g["means3D"][0, 0], e["mean_x"]) < 2e-3 and rel(g["means3D"][0, 1], e["mean_y"]) < 2e-3


def test_hip_tiny_disc_low_pass():
    W, H = 48, 48
    sc = analytic.axis_scene(W, H, [(4.0, 0.004, 0.003, 0.9, (0.2, 0.7, 0.5))], bg=(0.0, 0.0, 0.0))
    sc["means3D"][0, :2] = [0.013, -0.021]
    run = HipRun(scene_args(sc)).forward()
    rc, ro, rr, rxy = analytic.raycast_render(sc, sc["_rgb"])
    assert int(run.radii[0]) == rr[0] == 3
    _images_close(run, rc, ro, frac=1.0)


def test_hip_two_stacked_discs():
    W, H = 64, 64
    for o1 in (0.35, 0.85):
        sc = analytic.axis_scene(W, H, [(2.5, 0.10, 0.10, o1, (1.0, 0.1, 0.1)), (6.0, 0.40, 0.30, 0.9, (0.1, 0.2, 1.0))])
        run = HipRun(scene_args(sc)).forward()
        ec, eo = analytic.stacked_discs(sc)
        _images_close(run, ec, eo, frac=0.9995)
        o = run.others.cpu().numpy()
        assert o[5][H // 2, W // 2] == np.float32(6.0 if o1 < 0.5 else 2.5)
        assert o[6][H // 2, W // 2] > 1e-4
