"""Side legs of bench.py: forward-only Msplats/s @1080p and the CPU baseline (the oracle's fp32 port)."""
import os
import time

import numpy as np


def fwd_1080p(dev, name="1080p_1M", iters=20, warmup=5):
    import torch
    import synthetic
    from diff_surfel_rasterization import GaussianRasterizationSettings, GaussianRasterizer
    P, W, H, zf = synthetic.CONFIGS[name]
    sc = synthetic.make_scene(P, W, H, seed=0, z_far=zf)
    t = lambda x: torch.as_tensor(np.ascontiguousarray(x)).to(dev)
    rs = GaussianRasterizationSettings(image_height=H, image_width=W, tanfovx=sc["tanfovx"], tanfovy=sc["tanfovy"], bg=t(sc["bg"]),
                                       scale_modifier=1.0, viewmatrix=t(sc["viewmatrix"]), projmatrix=t(sc["projmatrix"]),
                                       sh_degree=3, campos=t(sc["campos"]), prefiltered=False, debug=False)
    rast = GaussianRasterizer(raster_settings=rs)
    means3D, shs, opac, scales, rots = (t(sc[k]) for k in ("means3D", "shs", "opacities", "scales", "rotations"))
    means2D = torch.zeros_like(means3D)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ms = []
    with torch.no_grad():
        for i in range(warmup + iters):
            e0.record()
            rast(means3D=means3D, means2D=means2D, shs=shs, colors_precomp=None, opacities=opac, scales=scales, rotations=rots,
                 cov3D_precomp=None)
            e1.record()
            torch.cuda.synchronize()
            if i >= warmup:
                ms.append(e0.elapsed_time(e1))
    import diff_surfel_rasterization
    med = float(np.median(ms))
    return {"workload": "%d surfels, %dx%d, forward only (preprocess..blend incl. sort)" % (P, W, H), "ms_median": round(med, 4),
            "Msplats_per_s": round(P / (med * 1e-3) / 1e6, 2), "instances_R": int(diff_surfel_rasterization.last_num_rendered),
            "Minst_per_s": round(diff_surfel_rasterization.last_num_rendered / (med * 1e-3) / 1e6, 2)}


def cpu_baseline(workload="C2"):
    """Oracle fp32 port (oracle/surfel_oracle.c -DORACLE_F32, OpenMP) timed on the host: one fwd+bwd of the
    same synthetic workload.  Reported beside the GPU figure; it is a baseline, not a target."""
    import synthetic
    from oracle.surfel_oracle import Oracle
    sample = workload if workload in ("C1", "C2", "C3") else "C2"
    P, W, H, zf = synthetic.CONFIGS[sample]
    sc = synthetic.make_scene(P, W, H, seed=0, z_far=zf)
    o = Oracle("f32")
    rng = np.random.default_rng(0)
    gC = rng.normal(size=(3, H, W)).astype(np.float32); gO = rng.normal(size=(7, H, W)).astype(np.float32)
    # bounded sample: whole fwd+bwd passes of the workload, repeated until ~10 s of CPU work (at most 20 passes)
    tf = tb = 0.0
    passes = 0
    while passes < 20 and (tf + tb) < 10.0:
        t0 = time.perf_counter()
        R, col, oth, radii, st = o.rasterize_forward(sc["bg"], sc["means3D"], None, sc["opacities"], sc["scales"], sc["rotations"], 1.0,
                                                     None, sc["viewmatrix"], sc["projmatrix"], sc["tanfovx"], sc["tanfovy"], H, W,
                                                     sc["shs"], 3, sc["campos"])
        t1 = time.perf_counter()
        o.rasterize_backward(st, gC, gO)
        t2 = time.perf_counter()
        tf += t1 - t0; tb += t2 - t1; passes += 1
    cores = int(os.environ.get("OMP_NUM_THREADS", os.cpu_count() or 1))
    return {"value": round(passes / (tf + tb), 4), "unit": "view-iters/s", "cores": cores, "kind": "port",
            "sample": "%s-synthetic (%d surfels, %dx%d): %d fwd+bwd passes of the fp32 OpenMP oracle port, %.1f s of wall time; "
                      "per pass fwd %.3f s, bwd %.3f s" % (sample, P, W, H, passes, tf + tb, tf / passes, tb / passes),
            "fwd_bwd_Msplats_per_s": round(P * passes / (tf + tb) / 1e6, 4)}
