import os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "2d-gaussian-splatting_amd"), os.path.join(REPO, "tests")):
    sys.path.insert(0, p)
import surfel_native as n
import synthetic
from helpers import HipRun, scene_args
for (P, W, H, rad, opac) in [(6000, 96, 64, 30.0, 0.02), (6000, 32, 32, 30.0, 0.02), (6000, 32, 16, 30.0, 0.02)]:
    sc = synthetic.make_scene(P, W, H, seed=6, px_radius=rad, z_near=2.0, z_far=8.0)
    sc["opacities"] = np.full_like(sc["opacities"], opac)
    a = scene_args(sc)
    rng = np.random.default_rng(8)
    gC = rng.normal(size=(3, H, W)).astype(np.float32); gO = np.zeros((7, H, W), np.float32)
    cols = rng.uniform(0, 1, (P, 3)).astype(np.float32)
    res = {}
    for cull in (0, n.OPT_NO_CULL):
        run = HipRun(a, colors_precomp=cols, debug=cull).forward()
        img = run.color.cpu().numpy()
        for nm, flag in (("rows", n.OPT_BWD_ROWS), ("quad", n.OPT_BWD_QUAD)):
            run.debug = flag
            g1 = run.backward(gC, gO)["colors"]
            g2 = run.backward(gC, gO)["colors"]
            res[(nm, bool(cull))] = g1
            print("  %s nocull=%d R %d reproducible %s" % (nm, bool(cull), run.R, np.array_equal(g1, g2)))
        res[("img", bool(cull))] = img
    print("P %d %dx%d: images equal cull/nocull: %s" % (P, W, H, np.array_equal(res[("img", False)], res[("img", True)])))
    keys = [("rows", False), ("rows", True), ("quad", False), ("quad", True)]
    for i in range(4):
        for j in range(i + 1, 4):
            d = np.abs(res[keys[i]].astype(np.float64) - res[keys[j]])
            print("   %s vs %s: ndiff %d maxd %.3e" % (keys[i], keys[j], int((d > 0).sum()), d.max()))
