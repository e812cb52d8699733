"""Analytic anchors for the (unpinned) rasterizer oracle — TEST INFRASTRUCTURE.

Two statements of what the rasterizer must compute that share NO code and no formulation with oracle/surfel_oracle.c
(which follows the upstream homography form  k = x*Tw - Tu, l = y*Tw - Tv, p = k x l):

  * `raycast_render`  — geometric ray / plane intersection (paper arXiv 2403.17888 eq. 5-7 read literally: a pixel's ray
    meets the disc's plane at  t = (c.n)/(d.n),  (u, v) = ((t d - c).t_u / s_u, (t d - c).t_v / s_v) ), intrinsics from
    tan(fov) and the pixel-centre convention of /root/reference/gaussian_renderer/__init__.py:69-74, bounding box of the
    projected 3-sigma ellipse found NUMERICALLY (dense sampling of the circle u^2 + v^2 = 9), then a plain per-pixel
    front-to-back loop with the thresholds SURVEY.md §2.1 lists.
  * `disc_on_axis` / `two_stacked_discs` — closed forms for fronto-parallel discs on the optical axis
    (u = (x - cx) z / (f s_u), alpha profile, depth, normal (0,0,-1) alpha, T, distortion w1 w2 (m1 - m2)^2, median).

tests/test_analytic_cpu.py holds the fp64 oracle to these; tests/test_gpu_analytic.py holds the HIP path to them.
"""
import math

import numpy as np

NEAR_N, FAR_N = 0.2, 100.0
ALPHA_MIN, ALPHA_MAX, T_EPS = 1.0 / 255.0, 0.99, 1e-4
FILTER_SIGMA = 0.707106          # sqrt(2)/2 low-pass (paper eq. 11)
SH_C0 = 0.28209479177387814      # /root/reference/utils/sh_utils.py:26


def quat_to_rot(q):
    """(w, x, y, z) -> 3x3, /root/reference/utils/general_utils.py:78-100 (normalised first)."""
    q = np.asarray(q, np.float64)
    w, x, y, z = q / np.linalg.norm(q)
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                     [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])


def intrinsics(W, H, tanfovx, tanfovy):
    return W / (2.0 * tanfovx), H / (2.0 * tanfovy), (W - 1) / 2.0, (H - 1) / 2.0


def pixel_map(sc):
    """3x3 A (and a ~1e-7 residue b) with  (x_pix w, y_pix w, w) = p_view @ A + b : the pinhole the rasterizer is HANDED, i.e. recovered in fp64 from the
    float32 viewmatrix / projmatrix inputs (projection = inv(view) @ full_proj, row-vector convention of scene/cameras.py:56-58)
    and the pixel-centre rule x_pix = ((x_ndc + 1) W - 1) / 2 of gaussian_renderer/__init__.py:69-74.  Equals
    [[fx,0,0],[0,fy,0],[cx,cy,1]] from tan(fov) up to the float32 rounding of the matrices (checked)."""
    W, H = int(sc["W"]), int(sc["H"])
    V = np.asarray(sc["viewmatrix"], np.float64); F = np.asarray(sc["projmatrix"], np.float64)
    Pm = np.linalg.inv(V) @ F
    A4 = np.stack([Pm[:, 0] * (W / 2.0) + Pm[:, 3] * ((W - 1) / 2.0), Pm[:, 1] * (H / 2.0) + Pm[:, 3] * ((H - 1) / 2.0), Pm[:, 3]], 1)
    A, b = A4[:3], A4[3]          # b: float32 rounding residue of the matrices' translation rows (~1e-7), kept so the map is exact
    fx, fy, cx, cy = intrinsics(W, H, float(sc["tanfovx"]), float(sc["tanfovy"]))
    assert np.allclose(A, [[fx, 0, 0], [0, fy, 0], [cx, cy, 1]], rtol=1e-5, atol=1e-4) and np.abs(b).max() < 1e-4
    return A, b


def _tile_rect(cx, cy, r, gx, gy):
    """SURVEY.md §2.1 'tile rect' (3DGS getRect): truncating division, clamped to the grid."""
    x0 = min(gx, max(0, int((cx - r) / 16))); y0 = min(gy, max(0, int((cy - r) / 16)))
    x1 = min(gx, max(0, int((cx + r + 15) / 16))); y1 = min(gy, max(0, int((cy + r + 15) / 16)))
    return x0, y0, x1, y1


def raycast_render(sc, colors, n_theta=200_000):
    """sc: scene dict in the reference's vocabulary (means3D, scales, rotations, opacities, viewmatrix, tanfovx/y, W, H, bg);
    colors [P,3] = the surfels' RGB.  Returns (color[3,H,W], allmap[7,H,W], radii[P], centres[P,2]) in fp64."""
    W, H = int(sc["W"]), int(sc["H"])
    A, b = pixel_map(sc)
    V = np.asarray(sc["viewmatrix"], np.float64)            # world_view_transform = W2C^T (cameras.py:56)
    Rwc, t = V[:3, :3].T, V[3, :3]
    P = sc["means3D"].shape[0]
    mod = float(sc.get("scale_modifier", 1.0))
    gx, gy = (W + 15) // 16, (H + 15) // 16
    ys, xs = np.mgrid[0:H, 0:W].astype(np.float64)
    Ainv = np.linalg.inv(A)
    org = -b @ Ainv                                                                # ray origin (0 up to the float32 residue)
    dirs = np.stack([xs, ys, np.ones_like(xs)], -1) @ Ainv                         # [H,W,3] view-space ray of every pixel
    dirs = dirs / dirs[..., 2:3]                                                   # d_z = 1 -> org_z + t = view depth
    th = np.linspace(0.0, 2 * math.pi, n_theta, endpoint=False)
    circ = 3.0 * np.stack([np.cos(th), np.sin(th)], 1)                            # the 3-sigma circle in (u, v)
    discs = []
    radii = np.zeros(P, np.int32); centres = np.zeros((P, 2))
    for i in range(P):
        c = Rwc @ np.asarray(sc["means3D"][i], np.float64) + t
        if c[2] <= 0.2:
            continue
        Rq = Rwc @ quat_to_rot(sc["rotations"][i])
        su, sv = mod * float(sc["scales"][i][0]), mod * float(sc["scales"][i][1])
        tu, tv, n = Rq[:, 0], Rq[:, 1], Rq[:, 2]
        if -(c @ n) < 0:          # face the camera
            n = -n
        # plane of the disc = span(tu, tv); with a float32 (not exactly orthonormal) view rotation that is cross(tu, tv), not n
        npl = np.cross(tu, tv)
        # ... and (u, v) are coordinates in the (tu, tv) basis, read off with the dual basis (= tu, tv themselves when orthonormal)
        tud = np.cross(tv, npl); tud /= tu @ tud
        tvd = np.cross(npl, tu); tvd /= tv @ tvd
        # numeric bounding box of the projected 3-sigma ellipse
        pts = c[None] + circ[:, :1] * su * tu[None] + circ[:, 1:] * sv * tv[None]
        assert (pts[:, 2] > 1e-6).all(), "analytic scenes keep the 3-sigma ellipse in front of the camera"
        pp = pts @ A + b
        px = pp[:, 0] / pp[:, 2]; py = pp[:, 1] / pp[:, 2]
        bx, by = 0.5 * (px.max() + px.min()), 0.5 * (py.max() + py.min())
        ex, ey = 0.5 * (px.max() - px.min()), 0.5 * (py.max() - py.min())
        r = int(math.ceil(max(ex, ey, 3.0 * FILTER_SIGMA)))
        x0, y0, x1, y1 = _tile_rect(bx, by, r, gx, gy)
        if (x1 - x0) * (y1 - y0) == 0:
            continue
        radii[i] = r; centres[i] = (bx, by)
        discs.append((np.float32(c[2]), i, c, tud, tvd, n, npl, su, sv, (bx, by), (x0, y0, x1, y1)))
    discs.sort(key=lambda d: (d[0], d[1]))      # float32 view depth, then index (stable radix sort on depth bits)
    bg = np.asarray(sc["bg"], np.float64)
    T = np.ones((H, W)); C = np.zeros((3, H, W)); D = np.zeros((H, W)); N = np.zeros((3, H, W))
    M1 = np.zeros((H, W)); M2 = np.zeros((H, W)); dist = np.zeros((H, W)); med = np.zeros((H, W))
    done = np.zeros((H, W), bool)
    tile_x, tile_y = (xs // 16).astype(int), (ys // 16).astype(int)
    for _, i, c, tu, tv, n, npl, su, sv, (bx, by), (x0, y0, x1, y1) in discs:
        in_rect = (tile_x >= x0) & (tile_x < x1) & (tile_y >= y0) & (tile_y < y1)
        dn = dirs @ npl
        with np.errstate(divide="ignore", invalid="ignore"):
            tt = ((c - org) @ npl) / dn
            q = org + dirs * tt[..., None] - c
            tt = tt + org[2]
            u = (q @ tu) / su; v = (q @ tv) / sv
        rho3d = u * u + v * v
        rho2d = 2.0 * ((bx - xs) ** 2 + (by - ys) ** 2)
        use3d = rho3d <= rho2d
        depth = np.where(use3d, tt, c[2])
        alpha = np.minimum(ALPHA_MAX, float(np.ravel(sc["opacities"][i])[0]) * np.exp(-0.5 * np.minimum(rho3d, rho2d)))
        ok = in_rect & ~done & (dn != 0) & np.isfinite(rho3d) & (depth >= NEAR_N) & (alpha >= ALPHA_MIN)
        testT = T * (1 - alpha)
        stop = ok & (testT < T_EPS)
        done |= stop                      # the terminating surfel is not composited
        ok &= ~stop
        w = np.where(ok, alpha * T, 0.0)
        m = FAR_N / (FAR_N - NEAR_N) * (1 - NEAR_N / np.where(ok, depth, 1.0))
        dist += w * (m * m * (1 - T) + M2 - 2 * m * M1)
        D += w * depth_safe(depth, ok)
        M1 += w * m; M2 += w * m * m
        med = np.where(ok & (T > 0.5), depth, med)
        N += w[None] * n[:, None, None]
        C += w[None] * np.asarray(colors[i], np.float64)[:, None, None]
        T = np.where(ok, testT, T)
    color = C + T[None] * bg[:, None, None]
    allmap = np.stack([D, 1 - T, N[0], N[1], N[2], med, dist])
    return color, allmap, radii, centres


def depth_safe(depth, ok):
    return np.where(ok, depth, 0.0)


# ------------------------------------------------------------------------------------------------ closed forms
def axis_camera(W, H, focal_mult=1.2):
    """Camera at the origin looking down +z, identity rotation, matrices in the reference's layout (synthetic.look_at_camera)."""
    import synthetic
    return synthetic.look_at_camera(W, H, focal_mult=focal_mult)


def axis_scene(W, H, discs, bg=(0.1, 0.2, 0.3)):
    """Fronto-parallel discs on the optical axis: discs = [(z, su, sv, opacity, rgb)], identity rotation, centre (0, 0, z)."""
    cam = axis_camera(W, H)
    P = len(discs)
    sc = dict(means3D=np.array([[0.0, 0.0, d[0]] for d in discs], np.float32),
              scales=np.array([[d[1], d[2]] for d in discs], np.float32),
              rotations=np.tile(np.array([[1.0, 0, 0, 0]], np.float32), (P, 1)),
              opacities=np.array([[d[3]] for d in discs], np.float32),
              bg=np.array(bg, np.float32), scale_modifier=1.0, sh_degree=0)
    rgb = np.array([d[4] for d in discs], np.float64)
    shs = np.zeros((P, 16, 3), np.float32)
    shs[:, 0] = (rgb - 0.5) / SH_C0                     # colour = C0 * sh0 + 0.5 (sh_utils.py:57-65, __init__.py:90-91)
    sc["shs"] = shs
    sc.update(cam)
    # the float32 inputs the rasterizer actually sees
    sc["_rgb"] = SH_C0 * shs[:, 0].astype(np.float64) + 0.5
    return sc


def _axis_intrinsics(sc):
    """(fx, fy, cx, cy) of the identity-pose camera as its float32 matrices state them (= W/(2 tan fovx) ... to ~1e-7)."""
    A, b = pixel_map(sc)
    assert A[0, 1] == 0 and A[1, 0] == 0 and A[0, 2] == 0 and A[1, 2] == 0 and A[2, 2] == 1 and not b.any()
    return A[0, 0], A[1, 1], A[2, 0], A[2, 1]


def disc_alpha(sc, i):
    """alpha [H,W] of fronto-parallel on-axis disc i, the mask of pixels where it is composited in isolation, and rho3d<=rho2d."""
    W, H = int(sc["W"]), int(sc["H"])
    fx, fy, cx, cy = _axis_intrinsics(sc)
    z = float(sc["means3D"][i][2]); su, sv = (float(s) for s in sc["scales"][i]); o = float(sc["opacities"][i][0])
    ys, xs = np.mgrid[0:H, 0:W].astype(np.float64)
    u = (xs - cx) * z / (fx * su); v = (ys - cy) * z / (fy * sv)
    rho3d = u * u + v * v
    rho2d = 2.0 * ((xs - cx) ** 2 + (ys - cy) ** 2)
    alpha = np.minimum(ALPHA_MAX, o * np.exp(-0.5 * np.minimum(rho3d, rho2d)))
    r = int(math.ceil(max(3 * su * fx / z, 3 * sv * fy / z, 3 * FILTER_SIGMA)))
    gx, gy = (W + 15) // 16, (H + 15) // 16
    x0, y0, x1, y1 = _tile_rect(cx, cy, r, gx, gy)
    in_rect = (xs // 16 >= x0) & (xs // 16 < x1) & (ys // 16 >= y0) & (ys // 16 < y1)
    return alpha, in_rect & (alpha >= ALPHA_MIN), rho3d <= rho2d, r, (u, v)


def stacked_discs(sc):
    """Closed-form images of N fronto-parallel on-axis discs (sorted by z): returns (color, allmap)."""
    W, H = int(sc["W"]), int(sc["H"])
    order = np.argsort(sc["means3D"][:, 2], kind="stable")
    bg = sc["bg"].astype(np.float64)
    T = np.ones((H, W)); C = np.zeros((3, H, W)); D = np.zeros((H, W)); A = np.zeros((H, W)); Nz = np.zeros((H, W))
    med = np.zeros((H, W)); ws, ms = [], []
    for i in order:
        alpha, hit, _, _, _ = disc_alpha(sc, i)
        z = float(sc["means3D"][i][2])
        assert (T * (1 - alpha))[hit].min(initial=1.0) >= T_EPS, "closed form assumes no early termination"
        w = np.where(hit, alpha * T, 0.0)
        C += w[None] * sc["_rgb"][i][:, None, None]
        D += w * z; Nz -= w
        med = np.where(hit & (T > 0.5), z, med)
        T = np.where(hit, T * (1 - alpha), T)
        ws.append(w); ms.append(FAR_N / (FAR_N - NEAR_N) * (1 - NEAR_N / z))
    dist = np.zeros((H, W))
    for a in range(len(ws)):                    # distortion = sum_{i<j} w_i w_j (m_i - m_j)^2  (paper eq. 13)
        for b in range(a + 1, len(ws)):
            dist += ws[a] * ws[b] * (ms[a] - ms[b]) ** 2
    color = C + T[None] * bg[:, None, None]
    allmap = np.stack([D, 1 - T, np.zeros((H, W)), np.zeros((H, W)), Nz, med, dist])
    return color, allmap


def single_disc_grads(sc, gC):
    """Closed-form gradients of  L = sum(gC * color)  for ONE fronto-parallel on-axis disc (index 0):
    dL/dopacity, dL/d(SH dc coefficients), dL/d(scale u), dL/d(scale v), dL/d(mean x), dL/d(mean y)."""
    W, H = int(sc["W"]), int(sc["H"])
    fx, fy, cx, cy = _axis_intrinsics(sc)
    alpha, hit, use3d, _, (u, v) = disc_alpha(sc, 0)
    z = float(sc["means3D"][0][2]); su, sv = (float(s) for s in sc["scales"][0]); o = float(sc["opacities"][0][0])
    assert alpha[hit].max() < ALPHA_MAX, "closed form assumes the 0.99 clamp is inactive"
    ys, xs = np.mgrid[0:H, 0:W].astype(np.float64)
    G = alpha / o
    dL_dalpha = np.where(hit, ((sc["_rgb"][0] - sc["bg"].astype(np.float64))[:, None, None] * gC).sum(0), 0.0)
    g = {}
    g["opacity"] = (dL_dalpha * G).sum()
    g["sh_dc"] = SH_C0 * (np.where(hit, alpha, 0.0)[None] * gC).sum((1, 2))
    a3 = dL_dalpha * alpha * use3d          # d alpha / d rho = -alpha / 2 on either branch
    g["scale_u"] = (a3 * u * u / su).sum()
    g["scale_v"] = (a3 * v * v / sv).sum()
    # mean x: 3-D branch du/dX = -1/su; low-pass branch d rho2d / dX = 4 (xc - x) fx / z
    a2 = dL_dalpha * alpha * (~use3d)
    g["mean_x"] = (a3 * u / su).sum() + (-0.5 * a2 * 4.0 * (cx - xs) * fx / z).sum()
    g["mean_y"] = (a3 * v / sv).sum() + (-0.5 * a2 * 4.0 * (cy - ys) * fy / z).sum()
    return g
