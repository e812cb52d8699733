#!/usr/bin/env python
"""bench.py — throughput of the surfel-rasterizer hot path on MI355X (contract: see task statement).

A "step" = one forward + backward pass of the rasterizer (the reference's render() native call +
its autograd backward, /root/reference/train.py:69,90) over one synthetic view, through the product's
drop-in Python surface (diff_surfel_rasterization.GaussianRasterizer), inputs resident in HBM.
N > 1: view-parallel — every rank renders a different view of the same replicated surfel set, then
ONE RCCL all-reduce of the per-surfel gradient bucket (58 floats/surfel). Weak scaling.

Prints ONE JSON line on rank 0.  `value` = whole-job views/s (= train-iteration rasterizer rate at N=1).
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(REPO, "2d-gaussian-splatting_amd"))
sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0   # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.3 TB/s achievable)


def algorithmic_bytes(stage, P, V, R, W, H, n_pass):
    """SURVEY.md §8d per-launch ALGORITHMIC bytes of each stage (76-B record figure, not our 80-B layout)."""
    HW = W * H
    return {
        "preprocess_fwd": P * (232 + 87),
        "scan": P * 8,
        "emit_instances": R * 12,
        "radix_sort": R * 24 * n_pass,
        "tile_ranges": R * 8,
        "blend_fwd": R * (4 + 76) + HW * 60,
        "zero_grec": 0,
        "blend_bwd": R * (4 + 76) + HW * (60 + 40) + V * 18 * 4 * 2,
        "preprocess_bwd": P * (87 + 232 + 72) + P * (232 + 12),
    }.get(stage, 0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="C2", help="synthetic config name (synthetic.CONFIGS); C2 = BASELINE configs[1] shape")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-1080p", action="store_true")
    ap.add_argument("--no-train-iter", action="store_true")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist
    import synthetic
    import surfel_native
    from diff_surfel_rasterization import GaussianRasterizationSettings, GaussianRasterizer

    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (no CPU fallback for the rasterizer)")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    P, W, H, zf = synthetic.CONFIGS[args.workload]
    sc = synthetic.make_scene(P, W, H, seed=0, z_far=zf, view_index=rank)
    t = lambda x: torch.as_tensor(np.ascontiguousarray(x)).to(dev)
    rs = GaussianRasterizationSettings(image_height=H, image_width=W, tanfovx=sc["tanfovx"], tanfovy=sc["tanfovy"], bg=t(sc["bg"]),
                                       scale_modifier=1.0, viewmatrix=t(sc["viewmatrix"]), projmatrix=t(sc["projmatrix"]),
                                       sh_degree=3, campos=t(sc["campos"]), prefiltered=False, debug=3)   # 3 = HIP events around the dominant kernel only, no sync
    rast = GaussianRasterizer(raster_settings=rs)
    params = [t(sc[k]).requires_grad_(True) for k in ("means3D", "shs", "opacities", "scales", "rotations")]
    means3D, shs, opac, scales, rots = params
    g = torch.Generator(device="cpu").manual_seed(1234 + rank)
    gC = torch.randn((3, H, W), generator=g).to(dev); gO = torch.randn((7, H, W), generator=g).to(dev)
    import surfel_dist
    bucket = surfel_dist.GradBucket(P, dev) if world > 1 else None
    if bucket is not None:      # the backward writes its gradients straight into the all-reduce bucket
        import diff_surfel_rasterization as _dsr
        _dsr.set_grad_arena(bucket.arena())
    state = {}

    def step():
        means2D = torch.zeros_like(means3D, requires_grad=True)
        color, radii, allmap = rast(means3D=means3D, means2D=means2D, shs=shs, colors_precomp=None, opacities=opac,
                                    scales=scales, rotations=rots, cov3D_precomp=None)
        torch.autograd.backward([color, allmap], [gC, gO])
        if world > 1:
            bucket.all_reduce(average=True)      # ONE collective per step over the flat 232 B/surfel bucket
        state["radii"] = radii
        for p_ in params:
            p_.grad = None

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    surfel_native.collect_stage_times()     # drop warm-up events
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    dom_stage = surfel_native.collect_stage_times()        # {"blend_bwd": (total_ms, launches)} from the timed region itself
    # per-stage breakdown: a second, untimed pass of the same steps with every stage bracketed by events
    # (bracketing all ~9 stages costs ~10 us each, which would perturb the timed region by ~9 % at this size)
    rast.raster_settings = rs._replace(debug=2)
    for _ in range(max(5, args.steps // 2)):
        step()
    fence()
    stages = surfel_native.collect_stage_times()
    stages.update(dom_stage)
    if world > 1:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    # ---- workload statistics for the roofline (P, V, R, n_pass)
    V = int((state["radii"] > 0).sum().item())
    import diff_surfel_rasterization
    R = int(diff_surfel_rasterization.last_num_rendered)
    tiles = ((W + 15) // 16) * ((H + 15) // 16)
    n_pass = -(-(32 + max(1, (tiles - 1).bit_length())) // 8)

    out = None
    if rank == 0:
        ms_per_step = dt / args.steps * 1e3
        views_per_s = world * args.steps / dt
        per_kernel = {k: v[0] / v[1] for k, v in stages.items()}
        dom = max(per_kernel, key=per_kernel.get) if per_kernel else None
        roof = None
        if dom:
            B = algorithmic_bytes(dom, P, V, R, W, H, n_pass)
            ach = B / (per_kernel[dom] * 1e-3) / 1e9
            traffic = None
            tf = os.path.join(REPO, "profiles", "pmc_traffic.json")
            if os.path.exists(tf):
                try:
                    traffic = json.load(open(tf)).get(args.workload, {}).get(dom)
                except Exception:
                    traffic = None
            roof = {"bound": "hbm", "kernel": dom, "achieved": round(ach, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(ach / HBM_PEAK_GBS, 5), "traffic": traffic, "algorithmic_bytes": int(B),
                    "kernel_ms": round(per_kernel[dom], 4),
                    "all_kernels_ms": {k: round(v, 4) for k, v in per_kernel.items()},
                    "all_kernels_GBps": {k: round(algorithmic_bytes(k, P, V, R, W, H, n_pass) / (v * 1e-3) / 1e9, 1)
                                         for k, v in per_kernel.items() if v > 0}}
        out = {"metric": "train iters/sec (rasterizer fwd+bwd per view) + fwd Msplats/s @1080p", "value": round(views_per_s, 3),
               "unit": "view-iters/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
               "dtype": "f32", "data": "synthetic",
               "config": {"workload": "%s-synthetic: %d random surfels, %dx%d, sh_degree 3, rasterizer fwd+bwd, 1 view/GPU/step"
                                      % (args.workload, P, W, H), "P": P, "visible": V, "instances_R": R, "n_pass": n_pass,
                          "tiles": tiles, "parallelism": "view-parallel dp%d, 1 all-reduce of 232 B/surfel" % world},
               "fwd_bwd_Msplats_per_s": round(world * P * args.steps / dt / 1e6, 2), "roofline": roof}

    # ---- forward-only Msplats/s @1080p (BASELINE metric, N=1 leg only)
    if rank == 0 and world == 1 and not args.no_1080p:
        from helpers_bench import fwd_1080p
        out["fwd_1080p"] = fwd_1080p(dev)

    # ---- full training iteration (losses, regularisers, Adam around the rasterizer), N=1 leg only
    if rank == 0 and world == 1 and not args.no_train_iter:
        from helpers_bench import train_iter
        import diff_surfel_rasterization as _d
        _d.set_grad_arena(None)
        out["train_iter"] = train_iter(dev, args.workload if args.workload in ("C1", "C2", "C3", "C4") else "C2")
        _d.set_grad_arena(None)

    # ---- CPU baseline: the oracle's fp32 OpenMP port, same workload shape, rank 0 / N=1 only
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from helpers_bench import cpu_baseline
        out["cpu_baseline"] = cpu_baseline(args.workload)

    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
