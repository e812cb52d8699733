"""Diagnostic: HIP fp32 vs oracle fp64, next to oracle fp32 vs oracle fp64 (same maths, fp32 arithmetic on the CPU)."""
import sys, os
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "2d-gaussian-splatting_amd"), os.path.join(REPO, "tests")):
    sys.path.insert(0, p)
import numpy as np
import synthetic
from helpers import HipRun, cosine, frac_close, oracle_forward, scene_args
from oracle.surfel_oracle import Oracle

name = sys.argv[1] if len(sys.argv) > 1 else "C1"
sc = synthetic.make_config(name)
a = scene_args(sc)
run = HipRun(a).forward()
dk = run.depths()
o64, o32 = Oracle("f64"), Oracle("f32")
R, col, oth, radii, st = oracle_forward(o64, a, depth_key=dk)
R32, col32, oth32, radii32, st32 = oracle_forward(o32, a, depth_key=dk)
rng = np.random.default_rng(9)
gC = rng.normal(size=col.shape).astype(np.float32); gO = rng.normal(size=oth.shape).astype(np.float32)
g = run.backward(gC, gO)
og = o64.rasterize_backward(st, gC, gO)
og32 = o32.rasterize_backward(st32, gC, gO)
print("R hip/f64/f32", run.R, R, R32, "radii mismatch hip", (run.radii.cpu().numpy() != radii).mean(), "f32", (radii32 != radii).mean())
c = run.color.cpu().numpy(); o = run.others.cpu().numpy()
for nm, x, x32, ref in [("color", c, col32, col)] + [("others%d" % i, o[i], oth32[i], oth[i]) for i in range(7)]:
    print("%-9s hip frac %.5f maxabs %.2e | cpu-f32 frac %.5f maxabs %.2e" % (nm, frac_close(x, ref, 1e-4, 1e-4), np.abs(x - ref).max(),
          frac_close(x32, ref, 1e-4, 1e-4), np.abs(x32 - ref).max()))
for k, ref, r32 in [("means3D", og.dL_dmeans3D, og32.dL_dmeans3D), ("scales", og.dL_dscales, og32.dL_dscales), ("rots", og.dL_drots, og32.dL_drots),
               ("opacity", og.dL_dopacity, og32.dL_dopacity), ("sh", og.dL_dsh, og32.dL_dsh), ("means2D", og.dL_dmean2D, og32.dL_dmean2D)]:
    x = g[k].reshape(ref.shape); scale = np.abs(ref).mean()
    print("%-8s hip frac %.5f cos %.8f | cpu-f32 frac %.5f cos %.8f" % (k, frac_close(x, ref, 1e-4 * scale, 2e-3), cosine(x, ref),
          frac_close(r32, ref, 1e-4 * scale, 2e-3), cosine(r32, ref)))
