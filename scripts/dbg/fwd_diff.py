import sys, os
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (REPO, os.path.join(REPO, "2d-gaussian-splatting_amd"), os.path.join(REPO, "tests")):
    sys.path.insert(0, p)
import synthetic
import surfel_native as n
from helpers import HipRun, scene_args
import test_gpu_parity as T
lib = n.load()
for kind in sys.argv[1:] or ["huge_faint", "long_lists"]:
    if kind == "long_lists":
        sc = synthetic.make_scene(60000, 160, 128, seed=12, px_radius=6.0, z_near=1.0, z_far=9.0)
        sc["opacities"] = np.full_like(sc["opacities"], 0.015)
    else:
        sc = T._walk_scene(kind)
    a = scene_args(sc)
    res = []
    for pipe in (0, 1, 1):
        lib.surfel_set_option(b"fwd_pipe", pipe)
        run = HipRun(a).forward()
        gx, gy = (a["W"] + 15) // 16, (a["H"] + 15) // 16
        al = lambda v: (v + 255) // 256 * 256
        import diff_surfel_rasterization as dsr
        off = dsr.image_layout(a["W"], a["H"])[2]
        HW = a["W"] * a["H"]
        buf = run.ia.last()
        ranges = buf[:gx * gy * 8].view(run.torch.int32).view(-1, 2).cpu().numpy()
        last = buf[off + 12 * HW: off + 16 * HW].view(run.torch.int32).view(a["H"], a["W"]).cpu().numpy()
        res.append((run.color.cpu().numpy(), last, ranges))
    c0, l0, r0 = res[0]
    print(kind, "R", run.R, "list lengths", (r0[:, 1] - r0[:, 0]).min(), (r0[:, 1] - r0[:, 0]).max())
    for i in (1, 2):
        c1, l1, _ = res[i]
        d = np.abs(c0 - c1).max(0)
        bad = d > 0
        print("  run", i, "pixels differing", int(bad.sum()), "of", bad.size, "max diff", float(d.max()), "last differs", int((l0 != l1).sum()))
        if bad.any():
            ys, xs = np.nonzero(bad)
            tiles = sorted(set(zip((ys // 16).tolist(), (xs // 16).tolist())))
            print("   tiles", tiles[:12], "n =", [int(r0[ty * gx + tx, 1] - r0[ty * gx + tx, 0]) for ty, tx in tiles[:12]])
            y, x = ys[0], xs[0]
            print("   first", (y, x), c0[:, y, x], c1[:, y, x], "last", l0[y, x], l1[y, x])
            print("   last where differs: old", l0[bad][:10], "new", l1[bad][:10])
lib.surfel_set_option(b"fwd_pipe", 1)
