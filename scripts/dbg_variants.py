import os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "2d-gaussian-splatting_amd"), os.path.join(REPO, "tests")):
    sys.path.insert(0, p)
import surfel_native as n
import synthetic
from helpers import HipRun, scene_args, oracle_forward
from oracle.surfel_oracle import Oracle
o = Oracle("f64")
for P, rad, opac in [(60, 30.0, 0.02), (200, 30.0, 0.02), (1000, 30.0, 0.02), (3000, 30.0, 0.02), (9000, 30.0, 0.02), (9000, 30.0, None), (3000, 8.0, 0.02)]:
    sc = synthetic.make_scene(P, 96, 64, seed=6, px_radius=rad, z_near=2.0, z_far=8.0)
    if opac is not None:
        sc["opacities"] = np.full_like(sc["opacities"], opac)
    a = scene_args(sc)
    rng = np.random.default_rng(8)
    gC = rng.normal(size=(3, a["H"], a["W"])).astype(np.float32); gO = rng.normal(size=(7, a["H"], a["W"])).astype(np.float32)
    run = HipRun(a).forward()
    R, col, oth, radii, st = oracle_forward(o, a, depth_key=run.depths())
    og = o.rasterize_backward(st, gC, gO)
    res = {}
    for nm, flag in (("rows", n.OPT_BWD_ROWS), ("quad", n.OPT_BWD_QUAD)):
        run.debug = flag
        res[nm] = run.backward(gC, gO)
    ref = dict(means3D=og.dL_dmeans3D, opacity=og.dL_dopacity, scales=og.dL_dscales, sh=og.dL_dsh)
    line = "P %d rad %g opac %s R %d |" % (P, rad, opac, run.R)
    for k in ("means3D", "opacity", "scales", "sh"):
        d = res["rows"][k].astype(np.float64) - res["quad"][k]
        er = np.abs(res["rows"][k].reshape(ref[k].shape) - ref[k]).max() / np.abs(ref[k]).max()
        eq = np.abs(res["quad"][k].reshape(ref[k].shape) - ref[k]).max() / np.abs(ref[k]).max()
        line += " %s: ndiff %d maxd %.2e relerr rows %.2e quad %.2e |" % (k, int((d != 0).sum()), np.abs(d).max(), er, eq)
    print(line, flush=True)
