"""Seeded synthetic surfel scenes shared by tests/ and bench.py.

Pure numpy; builds the camera matrices with the reference's conventions re-stated here:
  * getProjectionMatrix            /root/reference/utils/graphics_utils.py:51-71
  * world_view_transform = W2C^T   /root/reference/scene/cameras.py:56
  * full_proj = view^T-convention product  /root/reference/scene/cameras.py:58
(tests/golden/make_golden.py checks these against the imported reference functions.)
This module is input generation, not an oracle: the product's bench may import it.
"""
import math

import numpy as np

CONFIGS = {
    # name: (P, W, H, z_far)
    "C1": (10_000, 256, 256, 12.0),
    "C2": (300_000, 800, 800, 12.0),
    "C3": (200_000, 800, 600, 12.0),
    "C4": (2_000_000, 1600, 1060, 12.0),
    "C5": (10_000_000, 3840, 2160, 60.0),
    "1080p_1M": (1_000_000, 1920, 1080, 12.0),
    "1080p_2M": (2_000_000, 1920, 1080, 12.0),
}
# median projected 1-sigma radius in px where it is not the default 4 px x W / 1920 (heavy-footprint stand-in for a trained scene:
# R / P ~ 10 tile instances per surfel instead of ~2)
PX_RADIUS = {"C2H": 7.0}
CONFIGS["C2H"] = (300_000, 800, 800, 12.0)


def projection_matrix(znear, zfar, fovx, fovy):
    ty, tx = math.tan(fovy / 2), math.tan(fovx / 2)
    top, right = ty * znear, tx * znear
    Pm = np.zeros((4, 4), np.float32)
    Pm[0, 0] = 2.0 * znear / (2 * right)
    Pm[1, 1] = 2.0 * znear / (2 * top)
    Pm[3, 2] = 1.0
    Pm[2, 2] = zfar / (zfar - znear)
    Pm[2, 3] = -(zfar * znear) / (zfar - znear)
    return Pm


def look_at_camera(W, H, focal_mult=1.2, Rcw=None, t=None, znear=0.01, zfar=100.0):
    """Returns dict(viewmatrix, projmatrix, campos, tanfovx, tanfovy) in the reference's layout."""
    f = focal_mult * W
    tanfovx = 0.5 * W / f
    tanfovy = 0.5 * H / f
    fovx, fovy = 2 * math.atan(tanfovx), 2 * math.atan(tanfovy)
    Rt = np.eye(4, dtype=np.float64)
    if Rcw is not None:
        Rt[:3, :3] = Rcw
    if t is not None:
        Rt[:3, 3] = t
    view = Rt.astype(np.float32).T                                  # world_view_transform (transposed)
    proj = projection_matrix(znear, zfar, fovx, fovy).T             # projection_matrix (transposed)
    full = (view.astype(np.float32) @ proj).astype(np.float32)      # full_proj_transform
    campos = np.linalg.inv(view.astype(np.float64))[3, :3].astype(np.float32)
    return dict(viewmatrix=np.ascontiguousarray(view), projmatrix=np.ascontiguousarray(full), campos=campos,
                tanfovx=tanfovx, tanfovy=tanfovy, W=W, H=H)


def _rot(axis, ang):
    axis = np.asarray(axis, np.float64); axis /= np.linalg.norm(axis)
    K = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
    return np.eye(3) + math.sin(ang) * K + (1 - math.cos(ang)) * K @ K


def _base_pose(tilt=True):
    if tilt:
        return _rot([0.3, 1.0, 0.2], 0.15), np.array([0.1, -0.05, 0.2])
    return np.eye(3), np.zeros(3)


def view_camera(W, H, view_index=0, tilt=True):
    """Camera of training view `view_index` of a make_scene() scene (view 0 = the camera the surfels are laid out for; the others
    add a small rotation + shift, so view-parallel ranks see different images of the SAME surfels)."""
    Rcw, t = _base_pose(tilt)
    if view_index:
        Rcw = _rot([0.1, 1.0, 0.0], 0.02 * view_index) @ Rcw
        t = t + np.array([0.05 * view_index, 0.0, 0.02 * view_index])
    return look_at_camera(W, H, Rcw=Rcw, t=t)


def make_scene(P, W, H, seed=0, z_near=2.0, z_far=12.0, px_radius=None, tilt=True, sh_degree=3, view_index=0):
    """Random surfels filling 110 % of the frustum slab z in [z_near, z_far] (≈9 % culled off-screen).
    Scales are chosen so the median projected 1-sigma radius is `px_radius` px (default: 4 px at 1080p,
    scaled with resolution). Camera is slightly rotated/translated so no matrix entry is trivially 0."""
    rng = np.random.default_rng(seed)
    Rcw, t = _base_pose(tilt)
    cam0 = look_at_camera(W, H, Rcw=Rcw, t=t)       # surfels are laid out in the base camera's frustum
    cam = view_camera(W, H, view_index, tilt)
    f = 1.2 * W
    if px_radius is None:
        px_radius = 4.0 * W / 1920.0 * 1.0
        px_radius = max(px_radius, 1.5)
    z = rng.uniform(z_near, z_far, P)
    xv = rng.uniform(-1.1, 1.1, P) * cam0["tanfovx"] * z
    yv = rng.uniform(-1.1, 1.1, P) * cam0["tanfovy"] * z
    pv = np.stack([xv, yv, z], 1)                                   # view space
    pw = (pv - t) @ Rcw                                             # world = R^T (p_view - t)
    s0 = px_radius * z / f                                          # world size for px_radius at depth z
    scales = np.exp(rng.normal(0.0, 0.6, (P, 2))) * s0[:, None]
    rots = rng.normal(size=(P, 4)); rots /= np.linalg.norm(rots, axis=1, keepdims=True)
    opac = 1.0 / (1.0 + np.exp(-rng.normal(0.0, 2.0, (P, 1))))
    M = 16
    sh = np.zeros((P, M, 3), np.float32)
    sh[:, 0] = rng.normal(0.0, 1.0, (P, 3))
    sh[:, 1:] = rng.normal(0.0, 0.1, (P, M - 1, 3))
    scene = dict(means3D=pw.astype(np.float32), scales=scales.astype(np.float32), rotations=rots.astype(np.float32),
                 opacities=opac.astype(np.float32), shs=sh, sh_degree=sh_degree,
                 bg=np.zeros(3, np.float32), scale_modifier=1.0)
    scene.update(cam)
    return scene


def make_config(name, seed=0):
    P, W, H, zf = CONFIGS[name]
    return make_scene(P, W, H, seed=seed, z_far=zf, px_radius=PX_RADIUS.get(name))
