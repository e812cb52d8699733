"""Per-step wall-clock (host, with synchronisation) of rasterizer forward / backward for a workload: python scripts/time_steps.py C5 [steps]"""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "2d-gaussian-splatting_amd")); sys.path.insert(0, REPO)
import numpy as np, torch, synthetic
from diff_surfel_rasterization import GaussianRasterizationSettings, GaussianRasterizer
wl = sys.argv[1] if len(sys.argv) > 1 else "C2"; steps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
dev = torch.device("cuda:0")
P, W, H, zf = synthetic.CONFIGS[wl]
sc = synthetic.make_scene(P, W, H, seed=0, z_far=zf)
t = lambda x: torch.as_tensor(np.ascontiguousarray(x)).to(dev)
rs = GaussianRasterizationSettings(image_height=H, image_width=W, tanfovx=sc["tanfovx"], tanfovy=sc["tanfovy"], bg=t(sc["bg"]), scale_modifier=1.0,
                                   viewmatrix=t(sc["viewmatrix"]), projmatrix=t(sc["projmatrix"]), sh_degree=3, campos=t(sc["campos"]), prefiltered=False, debug=0)
rast = GaussianRasterizer(rs)
params = [t(sc[k]).requires_grad_(True) for k in ("means3D", "shs", "opacities", "scales", "rotations")]
gC = torch.randn((3, H, W), device=dev); gO = torch.randn((7, H, W), device=dev)
for i in range(steps):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    m2 = torch.zeros_like(params[0], requires_grad=True)
    col, radii, allmap = rast(means3D=params[0], means2D=m2, shs=params[1], colors_precomp=None, opacities=params[2], scales=params[3], rotations=params[4], cov3D_precomp=None)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    torch.autograd.backward([col, allmap], [gC, gO])
    t3 = time.perf_counter(); torch.cuda.synchronize(); t4 = time.perf_counter()
    for p in params: p.grad = None
    del col, radii, allmap, m2
    torch.cuda.synchronize(); t5 = time.perf_counter()
    print("step %d: fwd call %.2f ms (+sync %.2f), bwd call %.2f ms (+sync %.2f), free %.2f ms | reserved %.1f GB alloc %.1f GB" % (
        i, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3, (t5 - t4) * 1e3, torch.cuda.memory_reserved() / 2**30, torch.cuda.memory_allocated() / 2**30))
