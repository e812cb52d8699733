#!/usr/bin/env python
"""bench.py — training-iteration throughput of the surfel hot path on MI355X (contract: see task statement).

A "step" = ONE FULL TRAINING ITERATION of the reference's loop (/root/reference/train.py:54-138) on one synthetic view per
GPU: rasterizer forward (preprocess / sort / blend) -> L1 + SSIM -> normal + distortion regularisers -> rasterizer backward
-> densification statistics -> [N > 1: RCCL all-reduce of the 40 B/surfel geometry gradients + all-gather of the 12 B/surfel/rank
colour gradients, SH gradients rebuilt locally] -> Adam step.
Everything runs through the product's drop-in surface (surfel_trainer.Trainer over diff_surfel_rasterization +
include/surfel_train.h kernels); inputs (parameters, target images, cameras) are resident in HBM before the timed region.
N > 1: view-parallel — every rank trains on a different view of the same replicated surfel set per step (weak scaling).

Prints ONE JSON line on rank 0.  `value` = whole-job training iterations (views) per second.
Side legs at N = 1: rasterizer-only fwd+bwd, forward-only Msplats/s @1080p (BASELINE metric), CPU baseline (oracle port).
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


def algorithmic_bytes(stage, P, V, R, W, H, n_pass, Rs=None):
    """SURVEY.md §8d per-launch ALGORITHMIC bytes of each stage (76-B record figure, not our 80-B layout).  Rs = STAGED instances
    (Σ over tiles of the deepest list position any pixel composited): what a blend pass has to read and the backward has to write —
    the instances behind a tile's saturation point are touched by neither, so charging all R of them to the blend stages made
    crowded frames (C5) print more than the HBM peak.  Binning stages move all R."""
    HW = W * H
    Rs = R if Rs is None else min(R, Rs)
    return {
        "preprocess_fwd": P * (232 + 87),
        "scan": P * 8,
        "emit_instances": R * 12,
        "radix_sort": R * 24 * n_pass,
        "tile_ranges": R * 8,
        "blend_fwd": Rs * (4 + 76) + HW * 60,
        "zero_grec": 0,
        "blend_bwd": Rs * (4 + 76) + HW * (60 + 40) + V * 18 * 4 * 2,
        "preprocess_bwd": P * (87 + 232 + 72) + P * (232 + 12),
    }.get(stage, 0)


def roofline_object(per_kernel, workload, P, V, R, Rs, W, H, n_pass):
    """The `roofline` object of one bench leg: dominant rasterizer kernel, its algorithmic bytes / its measured mean duration against
    the HBM peak, the committed PMC traffic of that kernel on that workload (if profiles/ holds one), the VALU-issue yardstick."""
    if not per_kernel:
        return None
    dom = max(per_kernel, key=per_kernel.get)
    B = algorithmic_bytes(dom, P, V, R, W, H, n_pass, Rs)
    ach = B / (per_kernel[dom] * 1e-3) / 1e9
    # HBM traffic of the dominant kernel: PMC counters cannot be read from inside the process, so this is the figure of the
    # committed rocprofv3 pass over this same command (scripts/profile_gpu.sh), labelled as such
    traffic, traffic_src, walk = None, None, None
    tf = os.path.join(REPO, "profiles", "pmc_traffic.json")
    if os.path.exists(tf):
        try:
            tj = json.load(open(tf))
            ent = tj.get(workload, {})
            traffic = ent.get(dom)
            walk = ent.get("_blend_bwd_walk")
            if traffic is not None:
                traffic_src = ("profiles/pmc_traffic.json (%s): separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over `python bench.py --workload %s`%s, "
                               "not read in this run" % (ent.get("_tag", tj.get("_tag", "committed")), workload,
                                                         (", blend_bwd walk forced to '%s'" % walk) if walk else ""))
        except Exception:
            traffic = None
    # second yardstick for the VALU-bound blend kernels: issued VALU wave-instructions per launch (SQ_INSTS_VALU from the
    # committed PMC pass) against the chip's fp32 vector issue peak: 157.3 TFLOP/s / 128 flop per wave-FMA (MI355X_MICROARCH.md)
    valu = None
    try:
        import glob
        # (the PMC file of the same profiling run the traffic figure came from — same code, same forced walk; else the newest)
        tag_file = os.path.join(REPO, "profiles", "%s_%s_pmc.json" % (json.load(open(tf)).get(workload, {}).get("_tag", "?"), workload)) if os.path.exists(tf) else ""
        pm = tag_file if os.path.exists(tag_file) else sorted(glob.glob(os.path.join(REPO, "profiles", "r*_%s_pmc.json" % workload)))[-1]
        pj = json.load(open(pm))
        n_inst = max([v.get("SQ_INSTS_VALU", 0.0) for k, v in pj.items() if k.startswith(dom)] or [0.0]) or None
        if n_inst:
            rate = n_inst / (per_kernel[dom] * 1e-3) / 1e9
            # measured_ceiling: what independent v_fma_f32 streams reach on this chip — 2.6 shader cycles per wave-instruction per SIMD at
            # the 2.1 GHz such a grid sustains = 830 G/s over 1024 SIMDs (scripts/issue_probe.hip, profiles/r05_issue_probe.md: one
            # workgroup per CU, slowest wave; rounds 2 - 4 quoted 711 from a probe with loop overhead).  The blend kernels' own mix (a
            # quarter DPP adds at 4.3 cycles, compares / selects / min / max, transcendentals at 8.2) issues at 2.9 cycles per instruction:
            # C2's blend_bwd sits at ~0.75 of the pure issue time of its own instructions (same file).
            valu = {"wave_insts_per_launch": int(n_inst), "achieved_Ginst_per_s": round(rate, 1), "peak_Ginst_per_s": 1228.9,
                    "frac": round(rate / 1228.9, 4), "measured_ceiling_Ginst_per_s": 830.0,
                    "frac_of_measured_ceiling": round(rate / 830.0, 4), "source": os.path.basename(pm)}
    except Exception:
        valu = None
    return {"bound": "hbm", "kernel": dom, "achieved": round(ach, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(ach / HBM_PEAK_GBS, 5), "traffic": traffic, "traffic_source": traffic_src, "algorithmic_bytes": int(B),
            "instances_R": int(R), "instances_staged": None if Rs is None else int(Rs),
            "kernel_ms": round(per_kernel[dom], 4), "valu_issue": valu,
            "all_kernels_ms": {k: round(v, 4) for k, v in per_kernel.items()},
            "all_kernels_GBps": {k: round(algorithmic_bytes(k, P, V, R, W, H, n_pass, Rs) / (v * 1e-3) / 1e9, 1)
                                 for k, v in per_kernel.items() if v > 0}}


def step_roofline(kernels_ms, ms_per_step, workload, P, V, R, Rs, W, H):
    """The whole iteration against the chip: sum of the ALGORITHMIC bytes of the rasterizer's stages + the training kernels' bytes
    (loss: 3 passes over the 10-channel maps + target; Adam: 28 B per parameter float over 58 floats per surfel) / ms_per_step / HBM peak,
    and the sum of the issued VALU wave-instructions of the committed PMC pass of this workload / step / the fp32 issue peak."""
    if not kernels_ms or not P:
        return None
    tiles = ((W + 15) // 16) * ((H + 15) // 16)
    n_pass = -(-(32 + max(1, (tiles - 1).bit_length())) // 8)
    ras = sum(algorithmic_bytes(k, P, V or P, R or 0, W, H, n_pass, Rs) for k in kernels_ms)
    train = 3 * (10 + 3) * 4 * W * H * 2 + 58 * 28 * P
    total = ras + train
    out = {"algorithmic_bytes_per_step": int(total), "of_which_rasterizer": int(ras), "achieved_GBps": round(total / (ms_per_step * 1e-3) / 1e9, 1),
           "frac_of_hbm_peak": round(total / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "sum_kernels_ms": round(sum(kernels_ms.values()), 4)}
    try:
        import glob
        pm = sorted(glob.glob(os.path.join(REPO, "profiles", "r*_%s_pmc.json" % workload)))[-1]
        pj = json.load(open(pm))
        # (per-launch means of the committed pass; every kernel of the iteration runs once per step except the tile sort's two passes;
        # kernels that are not part of the steady-state step — knn, the one-off activation — are left out)
        valu = sum(v.get("SQ_INSTS_VALU", 0.0) * (2.0 if "os_pass" in k else 1.0) for k, v in pj.items()
                   if isinstance(v, dict) and not any(x in k for x in ("knn", "activate_kernel", "mark_visible")))
        if valu > 0:
            out["valu_wave_insts_per_step"] = int(valu)
            out["valu_frac_of_issue_peak"] = round(valu / (ms_per_step * 1e-3) / 1e9 / 1228.9, 4)
            out["valu_source"] = os.path.basename(pm)
    except Exception:
        pass
    return out


def _respawn(n):
    """Re-execute this command line under torch.distributed.run with n ranks on this node; returns its exit code."""
    import socket
    import subprocess
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC only on this driver (RCCL / cross-process HIP memory)
    env.setdefault("OMP_NUM_THREADS", "8")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % n, "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def _rehearse_spawn(args):
    """Launch-path check that needs no GPU: rendezvous over gloo, one all-reduce, rank 0 prints what a real run would report."""
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo")
        t = torch.ones(1)
        dist.all_reduce(t)
        seen = int(t.item())
        rank = dist.get_rank()
        dist.barrier()
        dist.destroy_process_group()
    else:
        seen, rank = 1, 0
    if rank == 0:
        print(json.dumps({"rehearsal": True, "n_gpus": world, "ranks_in_all_reduce": seen, "requested_gpus": args.gpus}))
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="garden", help="'garden' (default, the headline: a TRAINED state of >= 2 M surfels at 1600x1060 = north_star's 1-GPU target shape, "
                                                         "BASELINE configs[3]) / 'trained' (54 k surfels, 800x800), or a synthetic config name (synthetic.CONFIGS: C2 = BASELINE "
                                                         "configs[1] shape, C4, C5, C2H)")
    ap.add_argument("--state", default=None, help="trained workloads: .ply cache of the trained model (loaded if present, written otherwise)")
    ap.add_argument("--legs", default="C2,C4,C2H,trained,C5", help="comma-separated side legs at N = 1 (any of C2, C4, C2H, C3, C5, trained, garden)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-1080p", action="store_true")
    ap.add_argument("--no-raster-only", action="store_true")
    ap.add_argument("--no-train-iter", action="store_true", help="(kept for older scripts; the timed step IS the training iteration)")
    ap.add_argument("--no-scale-leg", action="store_true", help="N > 1: skip the second weak-scaling line (C4 per GPU)")
    ap.add_argument("--no-legs", action="store_true", help="skip the extra N = 1 workload legs (C4-synthetic, heavy-footprint C2H, trained state)")
    ap.add_argument("--no-full-train", action="store_true", help="skip the config-3 full-train leg (30 000 iterations of the reference schedule, ~30 - 60 s)")
    ap.add_argument("--no-sweep", action="store_true", help="skip the 1080p forward sweep over P = 0.3 ... 10 M surfels")
    ap.add_argument("--quick", action="store_true", help="headline leg only (= --no-legs --no-cpu-baseline --no-1080p --no-raster-only)")
    ap.add_argument("--sharding", choices=("views", "bands"), default="views",
                    help="N > 1: 'views' = one view per GPU per iteration (weak scaling, default); 'bands' = tile-band sharding of ONE view "
                         "per iteration across the GPUs (strong scaling; BASELINE config 5: --workload C5)")
    ap.add_argument("--rehearse-spawn", action="store_true",
                    help="only exercise the launch path: spawn / rendezvous (gloo, no GPU needed), one all-reduce, print n_gpus and exit")
    args = ap.parse_args()

    # ---- launch contract: `python bench.py --gpus N` run DIRECTLY must time N ranks.  Without a torch.distributed.run
    # environment this process re-executes itself under it (one rank per GPU, rendezvous on 127.0.0.1) and relays the result.
    if args.quick:
        args.no_legs = args.no_cpu_baseline = args.no_1080p = args.no_raster_only = args.no_full_train = args.no_sweep = True
    if "WORLD_SIZE" not in os.environ:
        if args.gpus > 1:
            raise SystemExit(_respawn(args.gpus))
    elif int(os.environ["WORLD_SIZE"]) != args.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher started WORLD_SIZE=%s ranks" % (args.gpus, os.environ["WORLD_SIZE"]))
    if args.rehearse_spawn:
        return _rehearse_spawn(args)

    import torch
    import torch.distributed as dist
    import synthetic
    import surfel_native

    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (no CPU fallback for the rasterizer)")
    backend = os.environ.get("SURFEL_DIST_BACKEND", "nccl")     # "gloo": rehearsal of the N>1 path with all ranks on one GPU
    if backend != "nccl":
        local = local % torch.cuda.device_count()
    elif world > torch.cuda.device_count():
        raise SystemExit("bench.py: --gpus %d needs %d HIP devices on this node, found %d (SURFEL_DIST_BACKEND=gloo rehearses the "
                         "N > 1 path with all ranks on the devices present)" % (world, world, torch.cuda.device_count()))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    from helpers_bench import TRAINED_PRESETS, make_trainer, reseed_views, snapshot_for_cpu, trained_trainer, window_census
    n_views = max(8, world)
    trained_info = None
    if args.workload in TRAINED_PRESETS:
        tr, trained_info = trained_trainer(dev, args.workload, args.state)
        P, (W, H) = int(tr.model.P), TRAINED_PRESETS[args.workload]["res"]
        n_views = len(tr.cams)
    else:
        P, W, H, zf = synthetic.CONFIGS[args.workload]
        tr = make_trainer(dev, args.workload, n_views=n_views, sharding=args.sharding)      # Trainer picks up the process group
    # what this box sustains, in this process, BEFORE the set-up iterations (helpers_bench.box_probe; its host-side parts leave the GPU idle for
    # tens of milliseconds: placed between the set-up iterations and the warm-up it cost the 20-step window 3 % — clocks)
    probe = None
    if rank == 0:
        try:
            from helpers_bench import box_probe
            probe = box_probe(dev)
        except Exception as e:      # noqa: BLE001 — an extra must not cost the run its headline
            probe = {"error": repr(e)}
    # set-up iterations before the contract's W warm-up steps: allocator high-water marks, binning-path heuristic, clocks.  A capture's
    # views differ 2-3x in instance count, i.e. in buffer sizes: one pass over ALL of them (an epoch of the trainer's view stack), so that
    # no view of the timed window is the first of its size — a hipMalloc inside the window is a multi-millisecond host stall
    PRIME = max(15, len(tr.cams) if trained_info else 0)
    if world == 1 and trained_info:
        reseed_views(tr, 4242)
    for _ in range(PRIME):
        tr.step()
    tr.pipe.debug = 3       # HIP events around the dominant kernel only (blend_bwd), resolved after the timed region, no sync
    tr.time_exchange = world > 1     # N > 1: events around the stream waits on the collectives -> exposed exchange time

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        tr.step()
    fence()
    if world == 1:
        # the timed window sees the SAME views in every run of this command (the views of a capture differ 2-3x in instance count): the
        # rocprofv3 passes of scripts/profile_gpu.sh, the census pass below and the bench legs then time and count the same frames
        reseed_views(tr)
    loss_first = float(tr.last["loss"])
    surfel_native.collect_stage_times()     # drop warm-up events
    tr.exchange_events = []
    ms0 = torch.cuda.memory_stats(dev)
    host_steps = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        h0 = time.perf_counter()
        tr.step()
        host_steps.append(time.perf_counter() - h0)
    fence()
    dt = time.perf_counter() - t0
    ms1 = torch.cuda.memory_stats(dev)
    # (diagnostics of the window, nothing is re-timed: device allocations inside it and the host's own time per step)
    timed_window = {"device_allocs": int(ms1.get("num_device_alloc", 0) - ms0.get("num_device_alloc", 0)),
                    "device_frees": int(ms1.get("num_device_free", 0) - ms0.get("num_device_free", 0)),
                    "host_ms_per_step_median": round(sorted(host_steps)[len(host_steps) // 2] * 1e3, 4), "host_ms_per_step_max": round(max(host_steps) * 1e3, 4)}
    dom_stage = surfel_native.collect_stage_times()        # {"blend_bwd": (total_ms, launches)} from the timed region itself
    exposed_ms = (sum(e0.elapsed_time(e1) for e0, e1 in tr.exchange_events) / args.steps) if tr.exchange_events else None
    tr.time_exchange = False
    wire = dict(tr.wire)
    coll_us = None
    if world > 1 and args.sharding == "views" and backend == "nccl":
        # the two collectives of the step on their real buffers, back to back and alone on the device: what the exchange costs when
        # NOTHING hides it (exposed_ms_per_step above is what was not hidden in the timed region)
        try:      # (an extra: must never cost the run its headline)
            from surfel_trainer import GEOM_FLOATS
            m = tr.model
            geo = m.grad[:GEOM_FLOATS * m.P].clone()
            gall = torch.empty((world, m.P, 3), dtype=torch.float32, device=dev)
            coll_us = {}
            for name, fn in (("all_reduce_geometry_40B_per_surfel", lambda: dist.all_reduce(geo, op=dist.ReduceOp.SUM)),
                             ("all_gather_colour_12B_per_surfel_per_rank", lambda: dist.all_gather_into_tensor(gall, m.gcol))):
                for _ in range(3):
                    fn()
                fence()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10):
                    fn()
                e1.record()
                fence()
                coll_us[name] = round(e0.elapsed_time(e1) * 100.0, 1)
            del geo, gall
        except Exception as e:      # noqa: BLE001
            coll_us = {"error": repr(e)}
    loss_last = float(tr.last["loss"])
    # per-stage breakdown of the rasterizer + the window's instance counts: a second, untimed pass over the SAME frames (same view order)
    # with every stage bracketed by events (bracketing all ~9 stages costs ~10 us each: not inside the timed region) and R / Rs / V read
    # back per frame -> the roofline's bytes and the kernel times belong to the same frames
    stages, census = window_census(tr, args.steps, W, H)
    fence()
    stages.update(dom_stage)      # the dominant kernel's duration: the one measured inside the timed window
    cpu_snapshot = snapshot_for_cpu(tr) if (rank == 0 and world == 1 and trained_info and not args.no_cpu_baseline) else None
    if world > 1:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    # ---- workload statistics for the roofline (P, V, R, n_pass): means over the window's frames
    V, R, Rs = census["V"], census["R"], census["Rs"]
    tiles = ((W + 15) // 16) * ((H + 15) // 16)
    n_pass = -(-(32 + max(1, (tiles - 1).bit_length())) // 8)

    out = None
    if rank == 0:
        ms_per_step = dt / args.steps * 1e3
        bands = world > 1 and args.sharding == "bands"
        iters_per_s = (1 if bands else world) * args.steps / dt
        per_kernel = {k: v[0] / v[1] for k, v in stages.items()}
        roof = roofline_object(per_kernel, args.workload, P, V, R, Rs, W, H, n_pass)
        if roof:
            roof["window"] = census
            if world == 1:
                try:      # (an extra: what a useful (pixel, surfel) pair costs in wave-instructions — PMC instructions / pairs counted live)
                    from helpers_bench import add_insts_per_pair, useful_pairs
                    add_insts_per_pair(roof, useful_pairs(tr))
                except Exception:      # noqa: BLE001
                    pass
        if roof and roof.get("valu_issue") and probe and probe.get("valu_Ginst_per_s_in_kernel_span"):
            # against what THIS box's VALUs issue on independent v_fma_f32 streams (box_probe), not the 157.3 TFLOP/s / 128 yardstick that
            # non-packed fp32 code cannot reach
            vi = roof["valu_issue"]
            vi["fma_ceiling_this_box_Ginst_per_s"] = probe["valu_Ginst_per_s_in_kernel_span"]
            vi["frac_of_fma_ceiling_this_box"] = round(vi["achieved_Ginst_per_s"] / probe["valu_Ginst_per_s_in_kernel_span"], 4)
        out = {"metric": "train iters/sec (full iteration: rasterizer fwd+bwd, L1+SSIM, normal+dist regularisers, Adam) + fwd Msplats/s @1080p",
               "value": round(iters_per_s, 3), "unit": "train-iters/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": round(ms_per_step, 4), "box_probe": probe, "higher_is_better": True, "scaling": "strong" if bands else "weak", "vs_baseline": None,
               "dtype": "f32", "data": "synthetic",
               "config": {"workload": (trained_info["workload"] + "; steady-state iteration, every loss term on") if trained_info else
                                      "%s-synthetic: %d random surfels, %dx%d, sh_degree 3, %d target views rendered from the unperturbed "
                                      "surfels, %s, lambda_dssim 0.2, lambda_normal 0.05, lambda_dist 1000, depth_ratio 1, "
                                      "Adam on 58 floats/surfel; %d set-up iterations before the warm-up" % (args.workload, P, W, H, n_views,
                                      "1 view/iteration split into row bands" if bands else "1 view/GPU/iteration", PRIME),
                          "P": P, "image": "%dx%d" % (W, H), "visible_mean": round(V, 1), "instances_R_mean": round(R, 1), "instances_staged_mean": round(Rs, 1),
                          "instances_R_range": [census["R_min"], census["R_max"]], "n_pass": n_pass, "tiles": tiles,
                          "trained_state": trained_info,
                          "parallelism": ("tile-band sharding of one view over %d GPUs (image bands all-gathered, same gradient exchange)" % world) if bands
                          else ("view-parallel dp%d; per iteration 1 all-reduce of 40 B/surfel (geometry gradients) + 1 all-gather of "
                                "12 B/surfel/rank (colour gradients -> SH gradients rebuilt locally)" % world)},
               "timed_window": timed_window, "world_size_seen": world, "backend": backend if world > 1 else None,
               "exchange": None if world == 1 else {"wire_bytes_per_step_per_gpu": wire, "exposed_ms_per_step": None if exposed_ms is None else round(exposed_ms, 4),
                                                     "exposed_frac_of_step": None if exposed_ms is None else round(exposed_ms / ms_per_step, 4),
                                                     "collective_us_standalone": coll_us, "early_gather_probe": getattr(tr, "early_gather_probe", None),
                                                     "note": "wire bytes = ring-algorithm bytes per GPU per step by collective; exposed = mean time the "
                                                             "compute stream spent waiting on the collectives (events around the stream waits)"},
               "loss_first": round(loss_first, 5), "loss_last": round(loss_last, 5),
               "train_Msplats_per_s": round((1 if bands else world) * P * args.steps / dt / 1e6, 2), "roofline": roof,
               "step_roofline": step_roofline(per_kernel, ms_per_step, args.workload, P, V, R, Rs, W, H)}

    if world > 1:
        dist.barrier()
    del tr
    import diff_surfel_rasterization as _d
    _d.set_grad_arena(None)
    torch.cuda.empty_cache()

    # ---- N > 1: a second weak-scaling line on BASELINE configs[3]'s per-GPU shape (C4: 2 M surfels, 1600x1060) — the headline C2 step
    # is 0.6 ms, where RCCL's launch latency alone is a visible fraction; same barrier / max-over-ranks timing
    if world > 1 and args.workload == "C2" and args.sharding == "views" and not args.no_scale_leg:
        try:      # (an extra: a failure here is reported, the headline line above stands; every rank takes the same path)
            tr = make_trainer(dev, "C4", n_views=max(8, world), sharding=args.sharding)
            for _ in range(8 + args.warmup):
                tr.step()
            fence()
            tr.time_exchange = True
            tr.exchange_events = []
            k4 = max(10, args.steps // 2)
            t0 = time.perf_counter()
            for _ in range(k4):
                tr.step()
            fence()
            dt4 = time.perf_counter() - t0
            ex4 = (sum(e0.elapsed_time(e1) for e0, e1 in tr.exchange_events) / k4) if tr.exchange_events else None
            tt = torch.tensor([dt4], device=dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt4 = float(tt.item())
            if rank == 0:
                out["scale_leg_C4"] = {"workload": "C4-synthetic: 2 000 000 random surfels, 1600x1060, full iteration, 1 view/GPU/iteration (weak scaling)",
                                       "value": round(world * k4 / dt4, 3), "unit": "train-iters/s", "ms_per_step": round(dt4 / k4 * 1e3, 4), "steps": k4,
                                       "exposed_ms_per_step": None if ex4 is None else round(ex4, 4), "wire_bytes_per_step_per_gpu": dict(tr.wire),
                                       "early_gather_probe": getattr(tr, "early_gather_probe", None)}
        except Exception as e:      # noqa: BLE001
            if rank == 0:
                out["scale_leg_C4"] = {"error": repr(e)}
        dist.barrier()
        tr = None
        _d.set_grad_arena(None)
        torch.cuda.empty_cache()

    # ---- rasterizer alone (the north-star hot path without loss / optimiser), N=1 leg only
    if rank == 0 and world == 1 and not args.no_raster_only:
        try:
            from helpers_bench import raster_fwd_bwd
            out["raster_fwd_bwd"] = raster_fwd_bwd(dev, args.workload)
        except Exception as e:      # noqa: BLE001
            out["raster_fwd_bwd"] = {"error": repr(e)}

    # ---- forward-only Msplats/s @1080p (BASELINE metric, N=1 leg only)
    if rank == 0 and world == 1 and not args.no_1080p:
        try:
            from helpers_bench import fwd_1080p
            out["fwd_1080p"] = fwd_1080p(dev)
        except Exception as e:      # noqa: BLE001
            out["fwd_1080p"] = {"error": repr(e)}

    if rank == 0 and world == 1 and not args.no_sweep:      # the metric's forward sweep at 1080p
        try:
            from helpers_bench import fwd_1080p_sweep
            out["fwd_1080p_sweep"] = fwd_1080p_sweep(dev)
        except Exception as e:      # noqa: BLE001 — an extra must not cost the run its headline
            out["fwd_1080p_sweep"] = {"error": repr(e)}

    # ---- more workloads through the same full iteration (N=1 only): BASELINE configs[3]'s per-GPU shape, a heavy-footprint
    # synthetic, and a TRAINED state (post-densification statistics) — VERDICT r1 weak #5
    if rank == 0 and world == 1 and not args.no_legs:
        from helpers_bench import config_leg, copy_bandwidth, trained_leg
        out["hbm_copy_probe"] = copy_bandwidth(dev)
        out["legs"] = {}
        for leg in [x for x in args.legs.split(",") if x]:
            try:      # (a side leg must never cost the run its headline line)
                if leg == args.workload:
                    continue      # (the headline itself)
                if leg in TRAINED_PRESETS:
                    out["legs"][leg] = trained_leg(dev, leg, steps=30 if leg == "trained" else 20, warmup=5)
                elif leg == "C5":      # BASELINE configs[4]'s per-frame shape on ONE GPU: 10 M surfels, 3840x2160 (~15 ms per iteration)
                    out["legs"][leg] = config_leg(dev, leg, steps=10, warmup=3, prime=5, walks=False)
                else:
                    out["legs"][leg] = config_leg(dev, leg, steps=30 if leg in ("C2", "C2H") else 20, warmup=5)
            except Exception as e:      # noqa: BLE001
                out["legs"][leg] = {"error": repr(e)}

    # ---- BASELINE configs[2] as a full train: the reference's 30 000-iteration schedule incl. densification and opacity resets
    if rank == 0 and world == 1 and not args.no_full_train:
        try:
            from helpers_bench import full_train_leg
            out["full_train_config3"] = full_train_leg(dev)
        except Exception as e:      # noqa: BLE001
            out["full_train_config3"] = {"error": repr(e)}

    # ---- CPU baselines, rank 0 / N=1 only: the oracle's fp32 OpenMP port of the rasterizer on the headline workload shape, and
    # BASELINE configs[0]: the dense pure-PyTorch rasterizer at C1
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        sys.path.insert(0, os.path.join(REPO, "tests"))
        from helpers_bench import cpu_baseline, cpu_dense_c1
        try:
            out["cpu_baseline"] = cpu_baseline(args.workload, snapshot=cpu_snapshot)
        except Exception as e:      # noqa: BLE001
            out["cpu_baseline"] = {"error": repr(e)}
        try:
            out["cpu_baseline_dense_torch_C1"] = cpu_dense_c1()
        except Exception as e:      # noqa: BLE001
            out["cpu_baseline_dense_torch_C1"] = {"error": repr(e)}

    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
