"""Pair coverage of a trained state by block granularity (run ON THE GPU BOX; numpy + the fp64 oracle on a sub-sample).
For a sample of surfels of one view: the pixels its alpha >= 1/255 footprint composites (no saturation: geometry only) against the
pixels of the blocks it touches, for 4x4 (the blend kernels' sub-tile), 4x2, 2x4 and 2x2 blocks — the lane utilisation a blend walk
could reach at that list granularity — and the number of blocks per surfel (list entries to build / visit).
    python scripts/block_stats.py [garden|trained|C4] [sample] [state.ply]
"""
import os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "2d-gaussian-splatting_amd")); sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))

wl = sys.argv[1] if len(sys.argv) > 1 else "garden"
ns = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
state = sys.argv[3] if len(sys.argv) > 3 else "/tmp/state_%s.ply" % wl

from oracle.surfel_oracle import Oracle
if wl in ("garden", "trained"):
    import torch
    import helpers_bench as HB
    tr, info = HB.trained_trainer(torch.device("cuda:0"), wl, state)
    sc = HB.snapshot_for_cpu(tr, view=0)
    del tr
else:
    import synthetic
    P, W, H, zf = synthetic.CONFIGS[wl]
    sc = synthetic.make_scene(P, W, H, seed=0, z_far=zf, px_radius=synthetic.PX_RADIUS.get(wl))
    sc = dict(sc); sc["P"], sc["W"], sc["H"] = P, W, H
P, W, H = sc["P"], sc["W"], sc["H"]
rng = np.random.default_rng(0)
pick = np.sort(rng.choice(P, size=min(ns, P), replace=False))
sub = {k: (np.ascontiguousarray(v[pick]) if isinstance(v, np.ndarray) and v.shape[:1] == (P,) else v) for k, v in sc.items()}
o = Oracle("f64")
R, col, oth, radii, st = o.rasterize_forward(sub["bg"], sub["means3D"], None, sub["opacities"], sub["scales"], sub["rotations"], 1.0, None,
                                             sub["viewmatrix"], sub["projmatrix"], sub["tanfovx"], sub["tanfovy"], H, W, sub["shs"], 3, sub["campos"])
T = st.transMat; xy = st.xy; opa = st.normal_opacity[:, 3]
vis = np.nonzero(radii > 0)[0]
grans = [(4, 4), (4, 2), (2, 4), (2, 2), (8, 8), (16, 16)]
tot_pix = 0
tot_blk = {g: 0 for g in grans}
hist = []
for i in vis:
    Tu, Tv, Tw = T[i, 0:3], T[i, 3:6], T[i, 6:9]
    r = int(radii[i]); cx, cy = xy[i]
    x0 = max(0, int((cx - r) // 16)) * 16; x1 = min(W, (int((cx + r + 15) // 16)) * 16)
    y0 = max(0, int((cy - r) // 16)) * 16; y1 = min(H, (int((cy + r + 15) // 16)) * 16)
    if x1 <= x0 or y1 <= y0:
        continue
    xs = np.arange(x0, x1); ys = np.arange(y0, y1)
    px, py = np.meshgrid(xs, ys)
    k = px[..., None] * Tw - Tu; l = py[..., None] * Tw - Tv
    p = np.cross(k, l)
    with np.errstate(all="ignore"):
        sx = p[..., 0] / p[..., 2]; sy = p[..., 1] / p[..., 2]
        rho3 = sx * sx + sy * sy
        rho2 = 2.0 * ((cx - px) ** 2 + (cy - py) ** 2)
        rho = np.minimum(rho3, rho2)
        depth = np.where(rho3 <= rho2, sx * Tw[0] + sy * Tw[1] + Tw[2], Tw[2])
        alpha = np.minimum(0.99, opa[i] * np.exp(-0.5 * rho))
    ok = (alpha >= 1.0 / 255) & (depth >= 0.2) & (p[..., 2] != 0)
    n = int(ok.sum())
    if n == 0:
        continue
    yy, xx = np.nonzero(ok)
    xx = xx + x0; yy = yy + y0
    tot_pix += n; hist.append(n)
    for (bw, bh) in grans:
        tot_blk[(bw, bh)] += len(np.unique((yy // bh) * 65536 + (xx // bw)))
m = len(hist)
print("workload %s: %d sampled surfels, %d visible with a footprint; pixels / surfel mean %.1f median %.0f p90 %.0f" % (wl, len(pick), m, tot_pix / m, np.median(hist), np.percentile(hist, 90)))
for (bw, bh) in grans:
    nb = tot_blk[(bw, bh)]
    print("  blocks %2dx%-2d : %.2f blocks / surfel, lane utilisation (geometry only) %.3f" % (bw, bh, nb / m, tot_pix / (nb * bw * bh)))
