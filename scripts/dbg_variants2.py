import os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "2d-gaussian-splatting_amd"), os.path.join(REPO, "tests")):
    sys.path.insert(0, p)
import surfel_native as n
import synthetic
from helpers import HipRun, scene_args
for (P, W, H, rad, opac) in [(5000, 16, 16, 30.0, 0.02), (4000, 16, 16, 30.0, 0.02), (5000, 16, 16, 30.0, 0.002), (9000, 96, 64, 30.0, 0.005), (6000, 96, 64, 30.0, 0.02)]:
    sc = synthetic.make_scene(P, W, H, seed=6, px_radius=rad, z_near=2.0, z_far=8.0)
    sc["opacities"] = np.full_like(sc["opacities"], opac)
    a = scene_args(sc)
    rng = np.random.default_rng(8)
    gC = rng.normal(size=(3, H, W)).astype(np.float32); gO = np.zeros((7, H, W), np.float32)
    cols = rng.uniform(0, 1, (P, 3)).astype(np.float32)
    run = HipRun(a, colors_precomp=cols).forward()
    res = {}
    for nm, flag in (("rows", n.OPT_BWD_ROWS), ("quad", n.OPT_BWD_QUAD)):
        run.debug = flag
        res[nm] = run.backward(gC, gO)
    d = np.abs(res["rows"]["colors"].astype(np.float64) - res["quad"]["colors"]).max(1)
    depth = run.depths()
    order = np.argsort(depth, kind="stable")
    rank = np.empty(P, int); rank[order] = np.arange(P)
    bad = np.nonzero(d > 0)[0]
    ncon = run.ia.last()   # image buffer
    print("P %d %dx%d opac %g R %d: %d surfels differ; ranks of differing: min %s max %s; maxd %.3e; max|g| %.3e" % (
        P, W, H, opac, run.R, bad.size, rank[bad].min() if bad.size else None, rank[bad].max() if bad.size else None, d.max(),
        np.abs(res["quad"]["colors"]).max()), flush=True)
    if bad.size:
        b = bad[np.argsort(rank[bad])][:8]
        for i in b:
            print("   surfel %d rank %d rows %s quad %s" % (i, rank[i], res["rows"]["colors"][i], res["quad"]["colors"][i]))
