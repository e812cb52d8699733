"""Dense fp64 PyTorch statement of the surfel rasterizer forward; backward = autograd.

TEST INFRASTRUCTURE ONLY.  Purpose: an *independent* second statement (the algorithm of the
"Surfel Rasterizer (Python)" notebook linked at /root/reference/README.md:3,7 — every pixel
against every surfel, global depth order, cumprod alpha blend) used to check that the hand-derived
backward in surfel_oracle.c is the gradient of its forward.  O(P*H*W): tiny scenes only.

In-tree sources followed:
  * homography T = (splat2world[:, [0,1,3]] @ world2pix[:, [0,1,3]])^T
        /root/reference/gaussian_renderer/__init__.py:64-75, scene/gaussian_model.py:27-33,
        utils/general_utils.py:78-110
  * SH colour  /root/reference/utils/sh_utils.py:57-112; +0.5 and clamp_min(0)
        /root/reference/gaussian_renderer/__init__.py:88-91
  * allmap channel meaning  /root/reference/gaussian_renderer/__init__.py:118-135
Where CUDA semantics differ from plain autograd the shim is explicit:
  clamp(0.99) is pass-through; quaternion normalisation factor is detached; the dual-visibility
  sign is a constant; masks/thresholds/termination are constants.
"""
import math

import torch

NEAR_N, FAR_N = 0.2, 100.0
FILTER_INV_SQUARE = 2.0
C0 = 0.28209479177387814
C1 = 0.4886025119029199
C2 = [1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396]
C3 = [-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154, -0.4570457994644658,
      1.445305721320277, -0.5900435899266435]


def build_rotation(q):
    # utils/general_utils.py:78-101, with the normalisation factor detached (see module doc)
    s = 1.0 / q.norm(dim=1, keepdim=True).detach()
    q = q * s
    r, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R = torch.stack([
        1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
        2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
        2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], dim=1).reshape(-1, 3, 3)
    return R


def trans_mat(means3D, scales, rotations, scale_modifier, projmatrix, W, H):
    """[P,9] = (Tu, Tv, Tw), i.e. cov3D_precomp of gaussian_renderer/__init__.py:75."""
    P = means3D.shape[0]
    R = build_rotation(rotations)
    S = torch.cat([scales * scale_modifier, torch.ones_like(scales[:, :1])], dim=1)
    RS = (R * S[:, None, :]).permute(0, 2, 1)            # gaussian_model.py:28
    trans = torch.zeros((P, 4, 4), dtype=means3D.dtype)
    trans[:, :3, :3] = RS
    trans[:, 3, :3] = means3D
    trans[:, 3, 3] = 1
    near, far = 0.01, 100.0
    ndc2pix = torch.tensor([[W / 2, 0, 0, (W - 1) / 2], [0, H / 2, 0, (H - 1) / 2], [0, 0, far - near, near],
                            [0, 0, 0, 1]], dtype=means3D.dtype).T
    world2pix = projmatrix @ ndc2pix
    T = (trans[:, [0, 1, 3]] @ world2pix[:, [0, 1, 3]]).permute(0, 2, 1).reshape(-1, 9)
    return T, R


def eval_sh(deg, sh, dirs):
    # sh [P,16,3]; utils/sh_utils.py:57-112
    x, y, z = dirs[:, 0:1], dirs[:, 1:2], dirs[:, 2:3]
    res = C0 * sh[:, 0]
    if deg > 0:
        res = res - C1 * y * sh[:, 1] + C1 * z * sh[:, 2] - C1 * x * sh[:, 3]
    if deg > 1:
        xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
        res = (res + C2[0] * xy * sh[:, 4] + C2[1] * yz * sh[:, 5] + C2[2] * (2 * zz - xx - yy) * sh[:, 6] +
               C2[3] * xz * sh[:, 7] + C2[4] * (xx - yy) * sh[:, 8])
    if deg > 2:
        res = (res + C3[0] * y * (3 * xx - yy) * sh[:, 9] + C3[1] * xy * z * sh[:, 10] +
               C3[2] * y * (4 * zz - xx - yy) * sh[:, 11] + C3[3] * z * (2 * zz - 3 * xx - 3 * yy) * sh[:, 12] +
               C3[4] * x * (4 * zz - xx - yy) * sh[:, 13] + C3[5] * z * (xx - yy) * sh[:, 14] +
               C3[6] * x * (xx - 3 * yy) * sh[:, 15])
    return res


def aabb_center(T):
    Tu, Tv, Tw = T[:, 0:3], T[:, 3:6], T[:, 6:9]
    t = torch.tensor([9.0, 9.0, -1.0], dtype=T.dtype)
    d = (t * Tw * Tw).sum(1, keepdim=True)
    f = t / d
    cx = (f * Tu * Tw).sum(1)
    cy = (f * Tv * Tw).sum(1)
    return torch.stack([cx, cy], 1)


def render_dense(means3D, scales, rotations, opacities, shs, colors_precomp, transMat_precomp, *, bg, viewmatrix,
                 projmatrix, campos, W, H, sh_degree, scale_modifier, radii, rects, depth_key, detach_center=False,
                 T_leaf=None, rows=None):
    """All tensors fp64 torch.  `radii` [P] int and `rects` [P,4] (x0,y0,x1,y1 tile rect) come from the C
    oracle's preprocess (non-differentiable culling/binning decisions); `depth_key` [P] is the float32
    depth used as sort key.  Returns color[3,H,W], allmap[7,H,W] (rows=(y0, y1): only those image rows, [.., y1-y0, W])."""
    dt = means3D.dtype
    vis = torch.as_tensor(radii) > 0
    idx = torch.nonzero(vis).squeeze(1)
    # stable global order: (depth_key, index)
    dk = torch.as_tensor(depth_key, dtype=torch.float64)[idx]
    order = idx[torch.argsort(dk, stable=True)]
    if transMat_precomp is None:
        T, R = trans_mat(means3D, scales, rotations, scale_modifier, projmatrix, W, H)
        nrm = R[:, :, 2] @ viewmatrix[:3, :3]
    else:
        T = transMat_precomp
        nrm = torch.tensor([0.0, 0.0, 1.0], dtype=dt).expand(means3D.shape[0], 3)
    if T_leaf is not None:
        T = T_leaf
    pview = means3D @ viewmatrix[:3, :3] + viewmatrix[3, :3]
    cosv = -(pview * nrm).sum(1)
    flip = torch.where(cosv > 0, 1.0, -1.0).detach()
    nrm = nrm * flip[:, None]
    xy = aabb_center(T.detach() if detach_center else T)
    if colors_precomp is None:
        d = means3D - campos
        d = d / d.norm(dim=1, keepdim=True)
        rgb = torch.clamp_min(eval_sh(sh_degree, shs, d) + 0.5, 0.0)
    else:
        rgb = colors_precomp

    T = T[order]; xy = xy[order]; nrm = nrm[order]; rgb = rgb[order]
    opa = opacities.reshape(-1)[order]
    rect = torch.as_tensor(rects)[order]
    y0, y1 = (0, H) if rows is None else rows
    ys, xs = torch.meshgrid(torch.arange(y0, y1, dtype=dt), torch.arange(W, dtype=dt), indexing="ij")
    H = y1 - y0
    px = xs.reshape(-1, 1); py = ys.reshape(-1, 1)                 # [N,1]
    tx = (px // 16).long(); ty = (py // 16).long()
    in_rect = (tx >= rect[:, 0]) & (tx < rect[:, 2]) & (ty >= rect[:, 1]) & (ty < rect[:, 3])   # [N,S]
    Tu, Tv, Tw = T[:, 0:3], T[:, 3:6], T[:, 6:9]
    k = px[:, :, None] * Tw[None] - Tu[None]                       # [N,S,3]
    l = py[:, :, None] * Tw[None] - Tv[None]
    p = torch.cross(k, l, dim=-1)
    pz_ok = p[..., 2] != 0
    pz = torch.where(pz_ok, p[..., 2], torch.ones_like(p[..., 2]))
    sx = p[..., 0] / pz; sy = p[..., 1] / pz
    rho3d = sx * sx + sy * sy
    dx = xy[None, :, 0] - px; dy = xy[None, :, 1] - py
    rho2d = FILTER_INV_SQUARE * (dx * dx + dy * dy)
    use3d = rho3d <= rho2d
    rho = torch.where(use3d, rho3d, rho2d)
    depth = torch.where(use3d, sx * Tw[None, :, 0] + sy * Tw[None, :, 1] + Tw[None, :, 2], Tw[None, :, 2].expand_as(sx))
    G = torch.exp(-0.5 * rho)
    a_raw = opa[None] * G
    alpha = a_raw + (torch.clamp(a_raw, max=0.99) - a_raw).detach()          # pass-through clamp
    valid = in_rect & pz_ok & (depth >= NEAR_N) & (alpha.detach() >= 1.0 / 255.0)
    a = torch.where(valid, alpha, torch.zeros_like(alpha))
    one_m = 1 - a
    T_incl = torch.cumprod(one_m, dim=1)
    T_excl = torch.cat([torch.ones_like(T_incl[:, :1]), T_incl[:, :-1]], dim=1)
    term = (valid & (T_incl.detach() < 1e-4))
    dead = torch.cummax(term.to(torch.int8), dim=1).values.bool()           # terminating surfel excluded too
    live = valid & ~dead
    w = torch.where(live, a * T_excl, torch.zeros_like(a))
    T_final = 1 - w.sum(1)
    m = FAR_N / (FAR_N - NEAR_N) * (1 - NEAR_N / torch.where(live, depth, torch.ones_like(depth)))
    m = torch.where(live, m, torch.zeros_like(m))
    dsafe = torch.where(live, depth, torch.zeros_like(depth))
    A_prev = torch.cumsum(w, 1) - w
    M1_prev = torch.cumsum(w * m, 1) - w * m
    M2_prev = torch.cumsum(w * m * m, 1) - w * m * m
    dist = (w * (m * m * A_prev + M2_prev - 2 * m * M1_prev)).sum(1)
    D = (w * dsafe).sum(1)
    N = (w[:, :, None] * nrm[None]).sum(1)
    Ccol = (w[:, :, None] * rgb[None]).sum(1) + T_final[:, None] * bg[None]
    med_mask = live & (T_excl.detach() > 0.5)
    S = w.shape[1]
    pos = torch.arange(1, S + 1)[None].expand_as(med_mask)
    med_idx = torch.where(med_mask, pos, torch.zeros_like(pos)).max(1).values    # 1-based, 0 = none
    med = torch.where(med_idx > 0, torch.gather(dsafe, 1, (med_idx - 1).clamp(min=0)[:, None]).squeeze(1),
                      torch.zeros_like(D))
    color = Ccol.T.reshape(3, H, W)
    allmap = torch.stack([D, 1 - T_final, N[:, 0], N[:, 1], N[:, 2], med, dist], 0).reshape(7, H, W)
    return color, allmap


def time_dense_forward(sc, st, max_seconds=25.0, rows_per_chunk=4, cores=None):
    """CPU baseline leg of bench.py (BASELINE configs[0]): forward of the dense rasterizer above in fp32 under no_grad on all host
    cores, over as many 4-row chunks of the image as fit into `max_seconds`, extrapolated linearly to the full image.
    sc = synthetic scene dict, st = the C oracle's preprocess state (radii / tile rects / depth keys: the non-differentiable
    binning decisions render_dense takes as inputs)."""
    import os
    import time
    import numpy as np
    W, H, P = int(sc["W"]), int(sc["H"]), sc["means3D"].shape[0]
    cores = cores or min(os.cpu_count() or 1, 32)      # intra-op threads actually used: more than ~32 slows these broadcast-heavy ops down
    torch.set_num_threads(cores)
    t = lambda x: torch.tensor(np.asarray(x, np.float32))
    gx, gy = (W + 15) // 16, (H + 15) // 16
    r = st.radii
    clampi = lambda v, hi: np.minimum(hi, np.maximum(0, v.astype(np.int64)))
    rects = np.stack([clampi(np.trunc((st.xy[:, 0] - r) / 16), gx), clampi(np.trunc((st.xy[:, 1] - r) / 16), gy),
                      clampi(np.trunc((st.xy[:, 0] + r + 15) / 16), gx), clampi(np.trunc((st.xy[:, 1] + r + 15) / 16), gy)], 1)
    kw = dict(bg=t(sc["bg"]), viewmatrix=t(sc["viewmatrix"]), projmatrix=t(sc["projmatrix"]), campos=t(sc["campos"]), W=W, H=H,
              sh_degree=int(sc["sh_degree"]), scale_modifier=1.0, radii=st.radii, rects=rects, depth_key=st.depths.astype(np.float32))
    args = [t(sc[k]) for k in ("means3D", "scales", "rotations", "opacities", "shs")]
    done_rows, spent = 0, 0.0
    with torch.no_grad():
        y = 0
        while y < H and (spent < max_seconds or done_rows == 0):
            t0 = time.perf_counter()
            render_dense(*args, None, None, rows=(y, min(H, y + rows_per_chunk)), **kw)
            spent += time.perf_counter() - t0
            done_rows += min(H, y + rows_per_chunk) - y
            y += rows_per_chunk * max(1, (H // rows_per_chunk) // 16)       # sample chunks spread over the image
    full = spent * H / done_rows
    return {"value": round(1.0 / full, 5), "unit": "forward views/s", "cores": cores, "kind": "port",
            "sample": "C1 (%d surfels, %dx%d), dense pure-PyTorch rasterizer (every surfel x every pixel, fp32, torch threads = %d): %d of %d "
                      "pixel rows in %.1f s, extrapolated linearly to the full image: %.1f s per forward view" % (P, W, H, cores, done_rows, H, spent, full),
            "Msplats_per_s": round(P / full / 1e6, 6)}
