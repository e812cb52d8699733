"""CPU tests (-m "not gpu"): pin the oracle against everything the reference states in-tree, and check
its hand-derived backward against an independent dense autograd statement."""
import numpy as np
import pytest


def _args_from_golden(g):
    return dict(bg=g["bg"], means3D=g["means3D"], opacities=g["opacities"], scales=g["scales"], rotations=g["rotations"],
                shs=g["shs"], viewmatrix=g["viewmatrix"], projmatrix=g["projmatrix"], campos=g["campos"],
                tanfovx=float(g["tanfovx"]), tanfovy=float(g["tanfovy"]), W=int(g["image_width"]), H=int(g["image_height"]),
                sh_degree=int(g["sh_degree"]), scale_modifier=1.0)


def test_transmat_matches_reference_python_twin(golden):
    """oracle preprocess homography == cov3D_precomp computed by the reference's own render()
    (/root/reference/gaussian_renderer/__init__.py:64-75) — fp32 reference vs fp64 oracle."""
    from helpers import oracle_forward
    from oracle.surfel_oracle import Oracle
    a = _args_from_golden(golden)
    _, _, _, radii, st = oracle_forward(Oracle("f64"), a)
    vis = radii > 0
    ref = golden["ref_cov3D_precomp"].astype(np.float64)
    assert vis.sum() > 400
    assert np.allclose(st.transMat[vis], ref[vis], rtol=2e-4, atol=2e-4)


def test_sh_colour_matches_reference_eval_sh(golden):
    """/root/reference/utils/sh_utils.py:57-112 + clamp_min(x+0.5, 0) for every active degree."""
    from helpers import oracle_forward
    from oracle.surfel_oracle import Oracle
    o = Oracle("f64")
    for deg in range(4):
        a = _args_from_golden(golden); a["sh_degree"] = deg
        _, _, _, radii, st = oracle_forward(o, a)
        vis = radii > 0
        assert np.allclose(st.rgb[vis], golden["ref_sh_rgb_deg%d" % deg][vis], atol=2e-6)
        assert np.array_equal(st.clamped[vis].astype(bool), (golden["ref_sh_rgb_deg%d" % deg][vis] == 0.0))


def test_camera_convention_matches_reference_camera(golden):
    """synthetic.look_at_camera == scene/cameras.py Camera (+ utils/graphics_utils.py) matrices."""
    import synthetic
    sc = synthetic.make_scene(512, 72, 56, seed=7, px_radius=4.0, z_near=1.0, z_far=6.0)
    assert np.array_equal(sc["viewmatrix"], golden["viewmatrix"])
    assert np.allclose(sc["projmatrix"], golden["projmatrix"], atol=1e-7)
    assert np.allclose(sc["campos"], golden["campos"], atol=1e-6)


def test_precomp_branch_self_consistency(golden):
    """Rendering with the reference's own python transMat as cov3D_precomp gives the same colour image as the
    native scale/rotation path (compute_cov3D_python True/False must agree)."""
    from helpers import oracle_forward
    from oracle.surfel_oracle import Oracle
    o = Oracle("f64")
    a = _args_from_golden(golden)
    R0, col0, oth0, rad0, _ = oracle_forward(o, a)
    R1, col1, oth1, rad1, _ = oracle_forward(o, a, transMat_precomp=golden["ref_cov3D_precomp"])
    assert abs(R0 - R1) <= 0.01 * R0
    assert np.mean(np.abs(col0 - col1) < 2e-3) > 0.99
    assert np.mean(np.abs(oth0[1] - oth1[1]) < 2e-3) > 0.99          # alpha
    # precomputed branch has no surfel normal: view normal is (0,0,-1) scaled by alpha
    assert np.allclose(oth1[4], -oth1[1], atol=1e-9) and np.allclose(oth1[2], 0)


def test_regression_vectors(golden):
    """oracle still reproduces its own committed fp64 outputs (guards accidental edits)."""
    from helpers import oracle_forward
    from oracle.surfel_oracle import Oracle
    o = Oracle("f64")
    a = _args_from_golden(golden)
    R, col, oth, radii, st = oracle_forward(o, a)
    assert R == int(golden["oracle_R"]) and np.array_equal(radii, golden["oracle_radii"])
    assert np.allclose(col, golden["oracle_color"], atol=1e-12) and np.allclose(oth, golden["oracle_others"], atol=1e-12)
    g = o.rasterize_backward(st, golden["grad_color"], golden["grad_others"])
    assert np.allclose(g.dL_dmeans3D, golden["oracle_dL_dmeans3D"], rtol=1e-9, atol=1e-9)
    assert np.allclose(g.dL_dsh, golden["oracle_dL_dsh"], rtol=1e-9, atol=1e-12)


def _rects(xy, r, W, H):
    gx, gy = (W + 15) // 16, (H + 15) // 16
    x0 = np.minimum(gx, np.maximum(0, np.trunc((xy[:, 0] - r) / 16).astype(int)))
    y0 = np.minimum(gy, np.maximum(0, np.trunc((xy[:, 1] - r) / 16).astype(int)))
    x1 = np.minimum(gx, np.maximum(0, np.trunc((xy[:, 0] + r + 15) / 16).astype(int)))
    y1 = np.minimum(gy, np.maximum(0, np.trunc((xy[:, 1] + r + 15) / 16).astype(int)))
    return np.stack([x0, y0, x1, y1], 1)


@pytest.mark.parametrize("seed,bgc", [(3, (0.3, 0.6, 0.1)), (4, (0.0, 0.0, 0.0))])
def test_backward_is_gradient_of_forward(seed, bgc):
    """C oracle forward == dense PyTorch forward to 1e-12, and its hand-derived backward == autograd of
    that dense forward (with the documented CUDA-semantics shims) to 1e-9 relative."""
    import torch
    import synthetic
    from helpers import oracle_forward, scene_args
    from oracle import dense_autograd as da
    from oracle.surfel_oracle import Oracle
    W, H, P = 56, 40, 300
    sc = synthetic.make_scene(P, W, H, seed=seed, px_radius=5.0, z_near=1.0, z_far=6.0)
    sc["bg"] = np.array(bgc, np.float32)
    a = scene_args(sc)
    o = Oracle("f64")
    R, col, oth, radii, st = oracle_forward(o, a)
    t = lambda x: torch.tensor(np.asarray(x, np.float64))
    leaves = {k: t(sc[k]).requires_grad_(True) for k in ["means3D", "scales", "rotations", "opacities", "shs"]}
    kw = dict(bg=t(sc["bg"]), viewmatrix=t(sc["viewmatrix"]), projmatrix=t(sc["projmatrix"]), campos=t(sc["campos"]), W=W, H=H,
              sh_degree=3, scale_modifier=1.0, radii=st.radii, rects=_rects(st.xy, st.radii, W, H),
              depth_key=st.depths.astype(np.float32))
    color, allmap = da.render_dense(leaves["means3D"], leaves["scales"], leaves["rotations"], leaves["opacities"], leaves["shs"],
                                    None, None, **kw)
    assert np.abs(color.detach().numpy() - col).max() < 1e-12
    assert np.abs(allmap.detach().numpy() - oth).max() < 1e-12
    rng = np.random.default_rng(5)
    gc = rng.normal(size=col.shape); go = rng.normal(size=oth.shape)
    loss = (color * t(gc)).sum() + (allmap * t(go)).sum()
    grads = torch.autograd.grad(loss, list(leaves.values()))
    g = o.rasterize_backward(st, gc, go)
    for k, ga, gb, tol in zip(leaves, grads, [g.dL_dmeans3D, g.dL_dscales, g.dL_drots, g.dL_dopacity, g.dL_dsh],
                              [1e-9, 1e-9, 1e-6, 1e-9, 1e-9]):     # rotations: input quats are unit only to fp32
        ga = ga.numpy().reshape(gb.shape)
        assert np.abs(ga - gb).max() <= tol * np.abs(ga).max(), k
    # densification statistic = dL/dTu.z * depth * W/2 of the blend-stage gradient (README.md:118: no low-pass term)
    T_leaf = t(st.transMat).requires_grad_(True)
    color2, allmap2 = da.render_dense(leaves["means3D"], leaves["scales"], leaves["rotations"], leaves["opacities"], leaves["shs"],
                                      None, None, detach_center=True, T_leaf=T_leaf, **kw)
    (gT,) = torch.autograd.grad((color2 * t(gc)).sum() + (allmap2 * t(go)).sum(), [T_leaf])
    gT = gT.numpy()
    vis = radii > 0
    stat_x = gT[:, 2] * st.transMat[:, 8] * 0.5 * W
    stat_y = gT[:, 5] * st.transMat[:, 8] * 0.5 * H
    assert np.allclose(g.dL_dmean2D[vis, 0], stat_x[vis], rtol=1e-8, atol=1e-9 * np.abs(stat_x).max())
    assert np.allclose(g.dL_dmean2D[vis, 1], stat_y[vis], rtol=1e-8, atol=1e-9 * np.abs(stat_y).max())
    assert np.all(g.dL_dmean2D[:, 2] == 0)


def test_f32_port_matches_f64():
    """the timed fp32 OpenMP port computes the same thing as the fp64 checker (fp32 tolerance)."""
    import synthetic
    from helpers import oracle_forward, scene_args
    from oracle.surfel_oracle import Oracle
    sc = synthetic.make_scene(2000, 128, 96, seed=1, px_radius=5.0)
    a = scene_args(sc)
    R, col, oth, radii, st = oracle_forward(Oracle("f64"), a)
    R32, col32, oth32, radii32, st32 = oracle_forward(Oracle("f32"), a)
    assert abs(R - R32) <= 2
    assert np.mean(np.abs(col - col32) < 1e-4) > 0.999


def test_radii_pin_uses_stage_one_alone():
    """determinacy.py pins radii with preprocess-only Monte-Carlo-arithmetic draws: the stage-1 entry reproduces the full forward's
    radii / extents exactly (fp64), a draw moves extents by fp32-sized amounts only, and the pinned set covers most surfels."""
    import synthetic
    import determinacy as D
    from helpers import oracle_forward, scene_args
    from oracle.surfel_oracle import Oracle
    sc = synthetic.make_scene(2000, 128, 96, seed=1, px_radius=5.0)
    a = scene_args(sc)
    args = (a["means3D"], None, a["opacities"], a["scales"], a["rotations"], a["scale_modifier"], None, a["viewmatrix"], a["projmatrix"],
            a["tanfovx"], a["tanfovy"], a["H"], a["W"], a["shs"], a["sh_degree"], a["campos"])
    R, col, oth, radii, st = oracle_forward(Oracle("f64"), a)
    r, e = Oracle("f64").preprocess_extents(*args)
    assert np.array_equal(r, radii) and np.array_equal(e, st.extent)
    om = Oracle("mca"); om.set_seed(7)
    rm, em = om.preprocess_extents(*args)
    assert np.abs(em - e).max() < 1e-2 and (rm != r).mean() < 0.01
    det = D.Determinacy(a, None, n_draws=1)
    assert det.radii_determined.mean() > 0.9 and np.array_equal(det.radii, radii)


def test_knn_oracle_known_answer():
    from oracle.surfel_oracle import Oracle
    pts = np.array([[0, 0, 0], [1, 0, 0], [0, 2, 0], [0, 0, 3], [10, 10, 10]], np.float32)
    d = Oracle("f64").knn_dist2(pts)
    assert np.isclose(d[0], (1 + 4 + 9) / 3)
    assert np.isclose(d[1], (1 + 5 + 10) / 3)


def test_oracle_edge_cases():
    import synthetic
    from helpers import oracle_forward, scene_args
    from oracle.surfel_oracle import Oracle
    o = Oracle("f64")
    sc = synthetic.make_scene(16, 40, 24, seed=0)
    a = scene_args(sc)
    a2 = dict(a); a2["means3D"] = a["means3D"] * np.array([1, 1, -1], np.float32) - np.array([0, 0, 5], np.float32)
    R, col, oth, radii, st = oracle_forward(o, a2)
    assert R == 0 and not radii.any() and np.all(oth == 0)
    assert np.allclose(col, np.broadcast_to(a["bg"][:, None, None], col.shape))
    a3 = dict(a)
    for k in ("means3D", "opacities", "scales", "rotations", "shs"):
        a3[k] = a[k][:0]
    R, col, oth, radii, st = oracle_forward(o, a3)
    assert R == 0
