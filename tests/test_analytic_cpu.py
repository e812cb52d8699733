"""External anchors for the unpinned rasterizer oracle: the fp64 oracle against
  (1) an independent geometric ray/plane renderer (tests/analytic.py::raycast_render),
  (2) closed forms for fronto-parallel discs on the optical axis (images AND gradients),
  (3) central finite differences of its own forward (the hand-derived backward is the gradient of the forward).
None of these share a formulation with oracle/surfel_oracle.c, which restates the upstream homography form.
"""
import numpy as np
import pytest

import analytic
from helpers import oracle_forward, scene_args


def _oracle():
    from oracle.surfel_oracle import Oracle
    return Oracle("f64")


@pytest.mark.parametrize("seed,tilt,P", [(1, True, 14), (2, False, 9), (4, True, 30), (6, True, 40)])
def test_oracle_matches_ray_plane_renderer(seed, tilt, P):
    import synthetic
    W, H = 72, 56
    sc = synthetic.make_scene(P, W, H, seed=seed, px_radius=7.0, z_near=2.0, z_far=7.0, tilt=tilt)
    rng = np.random.default_rng(seed)
    sc["bg"] = np.array([0.3, 0.1, 0.6], np.float32)
    sc["opacities"] = rng.uniform(0.05, 0.95, (P, 1)).astype(np.float32)
    cols = rng.uniform(0, 1, (P, 3)).astype(np.float32)
    a = scene_args(sc)
    R, col, oth, radii, st = oracle_forward(_oracle(), a, colors_precomp=cols)
    rc, ro, rr, rxy = analytic.raycast_render(sc, cols)
    assert np.array_equal(radii, rr), (radii, rr)
    vis = radii > 0
    assert vis.sum() >= P // 2
    assert np.abs(st.xy[vis] - rxy[vis]).max() < 1e-6          # AABB centre of the projected 3-sigma ellipse (numeric vs closed form)
    assert np.abs(st.depths[vis] - ((np.c_[sc["means3D"].astype(np.float64), np.ones(P)] @ sc["viewmatrix"].astype(np.float64))[vis, 2])).max() < 1e-12
    # images: every channel, every pixel (a pair sitting within 1e-8 of a threshold could flip; none does for these seeds)
    assert np.abs(col - rc).max() < 1e-7, np.abs(col - rc).max()
    for ch, nm in enumerate(["depth-sum", "alpha", "nx", "ny", "nz", "median", "distortion"]):
        assert np.abs(oth[ch] - ro[ch]).max() < 1e-7, (nm, np.abs(oth[ch] - ro[ch]).max())
    assert (ro[1] > 0.05).mean() > 0.2       # the scene actually covers a good part of the image


def test_disc_on_axis_known_answer():
    """One fronto-parallel disc on the optical axis: u = (x - cx) z / (f s_u), alpha = o exp(-min(u^2+v^2, 2|x-c|^2)/2),
    depth = z, view normal (0, 0, -1) alpha, median = z where it contributes, no distortion."""
    W, H = 64, 48
    sc = analytic.axis_scene(W, H, [(3.0, 0.12, 0.07, 0.8, (0.9, 0.4, 0.2))])
    a = scene_args(sc)
    o = _oracle()
    R, col, oth, radii, st = oracle_forward(o, a)
    alpha, hit, use3d, r, _ = analytic.disc_alpha(sc, 0)
    assert radii[0] == r == int(np.ceil(3 * 0.12 * 1.2 * W / 3.0))
    assert np.allclose(st.xy[0], [(W - 1) / 2, (H - 1) / 2], atol=1e-9)
    ec, eo = analytic.stacked_discs(sc)
    assert np.abs(col - ec).max() < 1e-12 and np.abs(oth - eo).max() < 1e-12
    assert hit.sum() > 150 and (~use3d & hit).sum() == 0       # a big disc: the low-pass branch never wins
    # gradients of L = sum(gC * color)
    rng = np.random.default_rng(0)
    gC = rng.normal(size=col.shape)
    g = o.rasterize_backward(st, gC, np.zeros_like(oth))
    e = analytic.single_disc_grads(sc, gC)
    assert np.isclose(g.dL_dopacity[0, 0], e["opacity"], rtol=1e-10)
    assert np.allclose(g.dL_dsh[0, 0], e["sh_dc"], rtol=1e-10)
    assert np.isclose(g.dL_dscales[0, 0], e["scale_u"], rtol=1e-9) and np.isclose(g.dL_dscales[0, 1], e["scale_v"], rtol=1e-9)
    assert np.isclose(g.dL_dmeans3D[0, 0], e["mean_x"], rtol=1e-9, atol=1e-12) and np.isclose(g.dL_dmeans3D[0, 1], e["mean_y"], rtol=1e-9, atol=1e-12)


def test_tiny_disc_low_pass_known_answer():
    """A sub-pixel disc: the screen-space low-pass min(rho3d, 2 |x - c|^2) carries the footprint (paper eq. 11) and its gradient
    reaches the mean through the projected centre only."""
    W, H = 48, 48
    sc = analytic.axis_scene(W, H, [(4.0, 0.004, 0.003, 0.9, (0.2, 0.7, 0.5))], bg=(0.0, 0.0, 0.0))
    sc["means3D"][0, :2] = [0.013, -0.021]       # off-axis by a fraction of a pixel: closed forms below use the ray-plane renderer
    a = scene_args(sc)
    o = _oracle()
    R, col, oth, radii, st = oracle_forward(o, a)
    rc, ro, rr, rxy = analytic.raycast_render(sc, sc["_rgb"])
    assert radii[0] == rr[0] == 3                 # ceil(3 * sqrt(2)/2): the filter radius, not the (sub-pixel) disc
    assert np.abs(col - rc).max() < 1e-9 and np.abs(oth - ro).max() < 1e-9
    assert (oth[1] > 0).sum() >= 9                # the footprint is several pixels although the disc is 0.1 px


def test_two_stacked_discs_known_answer():
    """Two fronto-parallel discs on the axis: T = (1-a1)(1-a2), C = a1 c1 + (1-a1) a2 c2 + T bg, distortion = w1 w2 (m1-m2)^2,
    median depth = the last disc composited while T > 0.5."""
    W, H = 64, 64
    for o1 in (0.35, 0.85):           # front disc below / above the alpha = 0.5 median switch
        sc = analytic.axis_scene(W, H, [(2.5, 0.10, 0.10, o1, (1.0, 0.1, 0.1)), (6.0, 0.40, 0.30, 0.9, (0.1, 0.2, 1.0))])
        a = scene_args(sc)
        R, col, oth, radii, st = oracle_forward(_oracle(), a)
        ec, eo = analytic.stacked_discs(sc)
        assert np.abs(col - ec).max() < 1e-12
        for ch in range(7):
            assert np.abs(oth[ch] - eo[ch]).max() < 1e-12, ch
        c = (H // 2, W // 2)
        assert oth[5][c] == (6.0 if o1 < 0.5 else 2.5)
        assert oth[6][c] > 1e-4            # the two discs are well separated in depth: visible distortion


def _fd_scene(seed=11, P=20, W=48, H=32):
    import synthetic
    sc = synthetic.make_scene(P, W, H, seed=seed, px_radius=6.0, z_near=2.0, z_far=5.0)
    rng = np.random.default_rng(seed)
    sc["opacities"] = rng.uniform(0.2, 0.85, (P, 1)).astype(np.float32)     # the pass-through 0.99 clamp stays inactive
    sc["bg"] = np.array([0.2, 0.4, 0.1], np.float32)
    return sc


def test_backward_matches_central_finite_differences():
    """dL/d(every input) of the fp64 oracle vs central differences of its own forward on a 20-surfel scene,
    L = sum(gC * color) + sum(gO * allmap) with the (piecewise constant) median-depth channel weighted too.
    Inputs are float32 in the ABI, so the probes x +- h are float32-exact and the true difference is the denominator; h is a
    few dozen float32 ulps (the forward itself is fp64), because the chance that a probe carries a (pixel, surfel) pair across
    one of the hard thresholds grows with h while the error such a jump causes grows with 1/h.
    Rotations: the kernel treats the incoming quaternion as unit (detached norm), so only the tangential part is compared."""
    o = _oracle()
    sc = _fd_scene()
    a = scene_args(sc)
    P, W, H = sc["means3D"].shape[0], a["W"], a["H"]
    rng = np.random.default_rng(3)
    gC = rng.normal(size=(3, H, W)); gO = rng.normal(size=(7, H, W))

    def loss(args):
        _, col, oth, _, st = oracle_forward(o, args)
        return float((gC * col).sum() + (gO * oth).sum()), st

    L0, st0 = loss(a)
    g = o.rasterize_backward(st0, gC, gO)
    visible = st0.radii > 0
    assert visible.sum() >= 15
    hand = dict(means3D=g.dL_dmeans3D, scales=g.dL_dscales, rotations=g.dL_drots, opacities=g.dL_dopacity, shs=g.dL_dsh)
    report = {}
    for name, rel_h in [("means3D", 4e-6), ("scales", 1e-5), ("rotations", 1e-5), ("opacities", 1e-5), ("shs", 1e-2)]:
        x0 = a[name].astype(np.float32)
        fd = np.zeros(x0.shape)
        it = np.ndindex(*x0.shape)
        for idx in it:
            if not visible[idx[0]]:
                continue
            if name == "shs" and idx[1] >= 16:
                continue
            h = rel_h * max(abs(float(x0[idx])), 0.05 if name != "means3D" else 1.0)
            xp, xm = x0.copy(), x0.copy()
            xp[idx] = np.float32(x0[idx] + h); xm[idx] = np.float32(x0[idx] - h)
            ap, am = dict(a), dict(a)
            ap[name], am[name] = xp, xm
            fd[idx] = (loss(ap)[0] - loss(am)[0]) / (float(xp[idx]) - float(xm[idx]))
        h_ = hand[name].reshape(x0.shape).astype(np.float64)
        if name == "rotations":           # compare tangential components (see docstring)
            q = x0.astype(np.float64); q /= np.linalg.norm(q, axis=1, keepdims=True)
            h_ = h_ - (h_ * q).sum(1, keepdims=True) * q
            fd = fd * np.linalg.norm(x0.astype(np.float64), axis=1, keepdims=True)      # d/dq_raw = (I - qq^T)/|q| d/dq_unit
            fd = fd - (fd * q).sum(1, keepdims=True) * q
        sel = np.broadcast_to(visible.reshape((-1,) + (1,) * (x0.ndim - 1)), x0.shape)
        scale = np.abs(h_[sel]).max()
        err = np.abs(fd - h_)[sel] / scale
        report[name] = (float(np.median(err)), float((err < 2e-5).mean()), float(err.max()))
        # a probe that carries a (pixel, surfel) pair across the 1/255, 1e-4, T > 0.5 or rho3d <= rho2d switch sees a jump, so a
        # few elements may be off; the bulk must agree to the O(h^2) truncation error
        assert np.median(err) < 2e-6, (name, report[name])
        assert (err < 2e-5).mean() >= 0.97, (name, report[name])
    print(report)
