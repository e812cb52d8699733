"""Footprint statistics of a synthetic workload (CPU, numpy): how many 8x8 wave tiles a surfel's alpha>=1/255 footprint
really touches vs. the tiles its bounding box touches, and the lane utilisation of a blend visit.
    python scripts/footprint_stats.py [workload] [sample]
"""
import os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "2d-gaussian-splatting_amd")); sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import synthetic
from helpers import scene_args, oracle_forward
from oracle.surfel_oracle import Oracle

wl = sys.argv[1] if len(sys.argv) > 1 else "C2"
ns = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
P, W, H, zf = synthetic.CONFIGS[wl]
sc = synthetic.make_scene(P, W, H, seed=0, z_far=zf)
a = scene_args(sc)
o = Oracle("f64")
prm_only = True
# preprocess only: reuse the oracle wrapper's arrays
import ctypes as C
from oracle import surfel_oracle as so
st = None
R, col, oth, radii, st = oracle_forward(o, a) if P <= 300000 else (None,) * 5
T = st.transMat; xy = st.xy; opa = st.normal_opacity[:, 3]
vis = np.nonzero(radii > 0)[0]
rng = np.random.default_rng(0)
sel = rng.choice(vis, size=min(ns, vis.size), replace=False)
tot_pix = tot_bbox8 = tot_exact8 = tot_ref16 = tot_bbox16 = tot_exact16 = tot_exact4 = 0
hist = []
for i in sel:
    Tu, Tv, Tw = T[i, 0:3], T[i, 3:6], T[i, 6:9]
    r = radii[i]; cx, cy = xy[i]
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
    tot_ref16 += ((x1 - x0 + 15) // 16) * ((y1 - y0 + 15) // 16)
    if n == 0:
        continue
    yy, xx = np.nonzero(ok)
    bx0, bx1, by0, by1 = xx.min() + x0, xx.max() + x0, yy.min() + y0, yy.max() + y0
    nb8 = (bx1 // 8 - bx0 // 8 + 1) * (by1 // 8 - by0 // 8 + 1)
    ne8 = len(set(zip((xx + x0) // 8, (yy + y0) // 8)))
    nb16 = (bx1 // 16 - bx0 // 16 + 1) * (by1 // 16 - by0 // 16 + 1)
    ne16 = len(set(zip((xx + x0) // 16, (yy + y0) // 16)))
    tot_exact4 += len(set(zip((xx + x0) // 4, (yy + y0) // 4)))
    tot_pix += n; tot_bbox8 += nb8; tot_exact8 += ne8; tot_bbox16 += nb16; tot_exact16 += ne16
    hist.append(n)
m = len(sel)
print("workload %s: sample %d visible surfels" % (wl, m))
print("  footprint pixels / surfel          : %.1f (median %.0f)" % (tot_pix / m, np.median(hist)))
print("  reference 16x16 rect tiles / surfel: %.2f" % (tot_ref16 / m))
print("  16x16 tiles, tight bbox / exact    : %.2f / %.2f" % (tot_bbox16 / m, tot_exact16 / m))
print("  8x8 tiles,  tight bbox / exact     : %.2f / %.2f" % (tot_bbox8 / m, tot_exact8 / m))
print("  4x4 sub-tiles, exact               : %.2f  (= %.2f wave passes of 4 rows; vs %.2f 8x8 visits)" % (tot_exact4 / m, tot_exact4 / m / 4.0, tot_exact8 / m))
print("  lane utilisation per 8x8 visit     : bbox %.3f, exact %.3f" % (tot_pix / (64.0 * tot_bbox8), tot_pix / (64.0 * tot_exact8)))
