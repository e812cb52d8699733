"""Side legs of bench.py: forward-only Msplats/s @1080p and the CPU baseline (the oracle's fp32 port)."""
import os
import time

import numpy as np


FWD_STAGES = ("preprocess_fwd", "scan", "emit_instances", "radix_sort", "tile_ranges", "blend_fwd")


def fwd_roofline(P, R, Rs, W, H, ms):
    """SURVEY 8d's B_fwd (preprocess 319 B/surfel + scan 8 B/surfel + 12 B/instance emitted + 24 B/instance/sort pass + 8 B/instance ranges
    + 80 B per blended instance + 60 B/pixel) over the measured forward time, against the HBM peak.  `achieved` charges the blend stage
    for the STAGED instances Rs (what a blend pass has to read); `survey_R` is the same with all R instances, as SURVEY 8d writes it."""
    from bench import HBM_PEAK_GBS, algorithmic_bytes
    tiles = ((W + 15) // 16) * ((H + 15) // 16)
    n_pass = -(-(32 + max(1, (tiles - 1).bit_length())) // 8)
    B = sum(algorithmic_bytes(k, P, P, R, W, H, n_pass, Rs) for k in FWD_STAGES)
    B_R = sum(algorithmic_bytes(k, P, P, R, W, H, n_pass, None) for k in FWD_STAGES)
    ach = B / (ms * 1e-3) / 1e9
    return {"bound": "hbm", "stage": "forward (preprocess .. blend incl. sort)", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(ach / HBM_PEAK_GBS, 4), "algorithmic_bytes": int(B), "survey_R": {"algorithmic_bytes": int(B_R), "achieved": round(B_R / (ms * 1e-3) / 1e9, 1),
                                                                                            "frac": round(B_R / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)},
            "P": int(P), "instances_R": int(R), "instances_staged": int(Rs), "n_pass": n_pass, "tiles": tiles, "ms": round(ms, 4)}


def _forward_only(dev, P, W, H, zf, iters, warmup):
    """Median ms of the forward under no_grad (the renderer's inference call, render.py:57 of the reference: no backward follows, so the
    forward leaves no tile stream behind), R and Rs of the frame."""
    import torch
    import synthetic
    import diff_surfel_rasterization as dsr
    from diff_surfel_rasterization import GaussianRasterizationSettings, GaussianRasterizer
    sc = synthetic.make_scene(int(P), W, H, seed=0, z_far=zf)
    t = lambda x: torch.as_tensor(np.ascontiguousarray(x)).to(dev)
    rs = GaussianRasterizationSettings(image_height=H, image_width=W, tanfovx=sc["tanfovx"], tanfovy=sc["tanfovy"], bg=t(sc["bg"]),
                                       scale_modifier=1.0, viewmatrix=t(sc["viewmatrix"]), projmatrix=t(sc["projmatrix"]),
                                       sh_degree=3, campos=t(sc["campos"]), prefiltered=False, debug=False)
    rast = GaussianRasterizer(raster_settings=rs)
    means3D, shs, opac, scales, rots = (t(sc[k]) for k in ("means3D", "shs", "opacities", "scales", "rotations"))
    del sc
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
    R = int(dsr.last_num_rendered)
    Rs = int(dsr.staged_instances(W, H))
    del means3D, shs, opac, scales, rots, means2D, rast
    torch.cuda.empty_cache()
    return float(np.median(ms)), R, Rs


def fwd_1080p(dev, name="1080p_1M", iters=20, warmup=5):
    import synthetic
    P, W, H, zf = synthetic.CONFIGS[name]
    med, R, Rs = _forward_only(dev, P, W, H, zf, iters, warmup)
    return {"workload": "%d surfels, %dx%d, forward only (preprocess..blend incl. sort), no_grad" % (P, W, H), "ms_median": round(med, 4),
            "Msplats_per_s": round(P / (med * 1e-3) / 1e6, 2), "instances_R": R, "instances_staged": Rs,
            "Minst_per_s": round(R / (med * 1e-3) / 1e6, 2), "roofline": fwd_roofline(P, R, Rs, W, H, med)}


CPU_THREADS = min(os.cpu_count() or 1, 32)      # BOTH CPU legs run on this many threads (the dense torch leg slows down beyond ~32)


def _omp_threads(n):
    """Thread count of the oracle's OpenMP regions (libgomp, the runtime oracle/liboracle_*.so links)."""
    import ctypes
    try:
        ctypes.CDLL("libgomp.so.1").omp_set_num_threads(int(n))
        return int(n)
    except OSError:
        return int(os.environ.get("OMP_NUM_THREADS", os.cpu_count() or 1))


def snapshot_for_cpu(tr, view=0):
    """Host copies of a Trainer's activated parameters and one of its cameras: the CPU baseline's sample of a trained workload."""
    import math
    m, cam = tr.model, tr.cams[view]
    P = int(m.P)
    f = lambda t: np.ascontiguousarray(t.detach().float().cpu().numpy())
    return dict(P=P, W=int(cam.image_width), H=int(cam.image_height), means3D=f(m._pv["xyz"]), shs=f(m._pv["sh"].view(P, 16, 3)), opacities=f(m._av["opacity"]),
                scales=f(m._av["scaling"]), rotations=f(m._av["rotation"]), viewmatrix=f(cam.world_view_transform), projmatrix=f(cam.full_proj_transform),
                tanfovx=math.tan(cam.FoVx * 0.5), tanfovy=math.tan(cam.FoVy * 0.5), campos=f(cam.camera_center), bg=np.zeros(3, np.float32))


def cpu_baseline(workload="C2", snapshot=None, budget_s=12.0):
    """Oracle fp32 port (oracle/surfel_oracle.c -DORACLE_F32, OpenMP) timed on the host: rasterizer fwd+bwd passes of the headline
    workload — the synthetic scene of a config name, or `snapshot` (snapshot_for_cpu: one view of the trained state the GPU leg
    timed).  Reported beside the GPU figure; it is a baseline, not a target."""
    import synthetic
    from oracle.surfel_oracle import Oracle
    if snapshot is not None:
        sc, P, W, H = snapshot, snapshot["P"], snapshot["W"], snapshot["H"]
        sample = "%s: one view of the trained state" % workload
    else:
        name = workload if workload in synthetic.CONFIGS else "C2"
        P, W, H, zf = synthetic.CONFIGS[name]
        sc = synthetic.make_scene(P, W, H, seed=0, z_far=zf, px_radius=synthetic.PX_RADIUS.get(name))
        sample = "%s-synthetic" % name
    o = Oracle("f32")
    cores = _omp_threads(CPU_THREADS)
    rng = np.random.default_rng(0)
    gC = rng.normal(size=(3, H, W)).astype(np.float32); gO = rng.normal(size=(7, H, W)).astype(np.float32)
    # bounded sample: whole fwd+bwd passes of the workload, repeated until ~budget_s of CPU work (at most 20 passes)
    tf = tb = 0.0
    passes = 0
    R = 0
    while passes < 20 and (tf + tb) < budget_s:
        t0 = time.perf_counter()
        R, col, oth, radii, st = o.rasterize_forward(sc["bg"], sc["means3D"], None, sc["opacities"], sc["scales"], sc["rotations"], 1.0,
                                                     None, sc["viewmatrix"], sc["projmatrix"], sc["tanfovx"], sc["tanfovy"], H, W,
                                                     sc["shs"], 3, sc["campos"])
        t1 = time.perf_counter()
        o.rasterize_backward(st, gC, gO)
        t2 = time.perf_counter()
        tf += t1 - t0; tb += t2 - t1; passes += 1
    return {"value": round(passes / (tf + tb), 4), "unit": "view-iters/s", "cores": cores, "kind": "port",
            "sample": "%s (%d surfels, %dx%d, %d tile instances): %d rasterizer fwd+bwd passes of the fp32 OpenMP oracle port on %d threads (the dominant part "
                      "of an iteration; loss and Adam are NOT included, which favours the CPU figure), %.1f s of wall time; "
                      "per pass fwd %.3f s, bwd %.3f s" % (sample, P, W, H, int(R), passes, cores, tf + tb, tf / passes, tb / passes),
            "fwd_bwd_Msplats_per_s": round(P * passes / (tf + tb) / 1e6, 4)}


def make_trainer(dev, workload="C2", n_views=8, regularizers=True, sharding="views"):
    """The bench's training job: the workload's synthetic surfels (synthetic.make_scene) seen from n_views nearby
    cameras; targets = renders of the unperturbed surfels; the trained model starts from perturbed parameters.  Every loss term
    is on (the DTU configuration: lambda_dist 1000, lambda_normal 0.05, depth_ratio 1) unless regularizers=False.  Densification
    is off (it runs every 100 iterations in the reference and is host-side torch indexing there as here)."""
    import math
    import torch
    import synthetic
    import surfel_model
    import surfel_trainer as TR
    from surfel_render import Camera
    P, W, H, zf = synthetic.CONFIGS[workload]
    sc0 = synthetic.make_scene(P, W, H, seed=0, z_far=zf, px_radius=synthetic.PX_RADIUS.get(workload))
    cams = []
    for k in range(n_views):
        sc = synthetic.view_camera(W, H, k)           # the same surfels seen from view k (no second 2 M-surfel generation per view)
        w2c = sc["viewmatrix"].T.astype(np.float64)
        cams.append(Camera(colmap_id=k, R=w2c[:3, :3].T, T=w2c[:3, 3], FoVx=2 * math.atan(sc["tanfovx"]), FoVy=2 * math.atan(sc["tanfovy"]),
                           image=torch.zeros(3, H, W), image_name="bench_%d" % k, uid=k, data_device=dev))
    t = lambda x: torch.as_tensor(np.ascontiguousarray(x))
    raw = dict(xyz=t(sc0["means3D"]), f_dc=t(sc0["shs"][:, :1]), f_rest=t(sc0["shs"][:, 1:]), opacity=torch.logit(t(sc0["opacities"]).clamp(1e-4, 1 - 1e-4)),
               scaling=torch.log(t(sc0["scales"])), rotation=t(sc0["rotations"]))
    gt = surfel_model.GaussianModel(3, device=dev)
    gt.set_parameters(**raw); gt.active_sh_degree = 3
    bg = torch.zeros(3, device=dev)
    TR.capture_views(gt, cams, bg)
    del gt
    g = torch.Generator().manual_seed(7)
    model = surfel_model.GaussianModel(3, device=dev)
    model.set_parameters(raw["xyz"] + 0.01 * torch.randn(raw["xyz"].shape, generator=g), raw["f_dc"] * 0.5, raw["f_rest"] * 0.0,
                         raw["opacity"] - 0.5, raw["scaling"] + 0.1 * torch.randn(raw["scaling"].shape, generator=g), raw["rotation"])
    model.active_sh_degree = 3
    model.spatial_lr_scale = TR.cameras_extent(cams) if n_views > 1 else 1.0
    return TR.Trainer(model, cams, _frozen_schedule(TR, regularizers), TR.pipeline_params(depth_ratio=1.0), sharding=sharding)


def _frozen_schedule(TR, regularizers=True):
    """Steady-state iteration: every loss term on, no densification / opacity reset inside the timed window."""
    far = 10 ** 9
    return TR.optimization_params(iterations=far, densify_from_iter=far, opacity_reset_interval=far, dist_from_iter=0 if regularizers else far,
                                  normal_from_iter=0 if regularizers else far, lambda_dist=1000.0, lambda_normal=0.05)


WINDOW_SEED = 20260922      # the view order of every timed window (bench legs, rocprofv3 passes): the same frames in every run of a command


def reseed_views(tr, seed=WINDOW_SEED):
    """Restart the Trainer's random view order (train.py:64-67's stack) from a fixed seed."""
    if hasattr(tr, "_rng") and hasattr(tr, "_stack"):
        tr._rng.seed(seed); tr._stack = []


def window_census(tr, steps, W, H, seed=WINDOW_SEED):
    """Replays the timed window's frames (same seed -> same views in the same order; the parameters have moved by `steps` Adam steps)
    with every rasterizer stage bracketed by events, and sums per FRAME what the roofline needs: instances R, staged instances Rs,
    visible surfels V.  Returns ({stage: (total_ms, launches)}, {"R": mean, "Rs": mean, "V": mean, "R_min": .., "R_max": ..}) — bytes and kernel times
    of a leg then belong to the same frames (the views of a trained capture differ 2 - 3x in instance count)."""
    import torch
    import surfel_native
    import diff_surfel_rasterization as dsr
    surfel_native.collect_stage_times()
    reseed_views(tr, seed)
    tr.pipe.debug = 2
    Rl, Rsl, Vl = [], [], []
    for _ in range(steps):
        tr.step()
        torch.cuda.synchronize()
        Rl.append(int(dsr.last_num_rendered)); Rsl.append(int(dsr.staged_instances(W, H))); Vl.append(int((tr.last["radii"] > 0).sum().item()))
    st = surfel_native.collect_stage_times()
    tr.pipe.debug = 0
    n = float(len(Rl))
    return st, {"R": sum(Rl) / n, "Rs": sum(Rsl) / n, "V": sum(Vl) / n, "R_min": min(Rl), "R_max": max(Rl), "frames": len(Rl)}


def useful_pairs(tr, frames=4, seed=WINDOW_SEED):
    """Mean composited (pixel, surfel) pairs per frame and the lane slots the backward walk issued for them, counted by the instrumented
    twin of the walk the library picks by itself (surfel_debug_set_blend_stats) over the first `frames` views of the timed window."""
    import torch
    import surfel_native
    lib = surfel_native.load()
    reseed_views(tr, seed)
    bs = torch.zeros(8, dtype=torch.int64, device=tr.model.device)
    lib.surfel_debug_set_blend_stats(surfel_native.ptr(bs))
    dbg = tr.pipe.debug
    tr.pipe.debug = 0
    try:
        for _ in range(frames):
            tr.step()
        torch.cuda.synchronize()
    finally:
        lib.surfel_debug_set_blend_stats(None)
        tr.pipe.debug = dbg
    sv = bs.cpu().numpy()
    return {"useful_pairs_per_frame": float(sv[1]) / frames, "useful_lane_frac": round(float(sv[1]) / max(1.0, float(sv[0])), 4), "frames": frames}


def add_insts_per_pair(roof, pairs):
    """roofline.valu_issue.wave_insts_per_useful_pair = the PMC pass's VALU wave-instructions per blend_bwd launch / the composited pairs
    per frame counted live: what a useful pair costs, lanes that composite nothing included."""
    if roof and roof.get("valu_issue") and pairs and pairs.get("useful_pairs_per_frame"):
        v = roof["valu_issue"]
        v["useful_pairs_per_frame"] = int(pairs["useful_pairs_per_frame"])
        v["useful_lane_frac"] = pairs["useful_lane_frac"]
        v["wave_insts_per_useful_pair"] = round(v["wave_insts_per_launch"] / pairs["useful_pairs_per_frame"], 3)
    return roof


def time_trainer(tr, steps, warmup, prime=15, workload=None, walks=True):
    """ms per full training iteration of an existing Trainer + the rasterizer's per-stage kernel times (second, untimed pass)."""
    import torch
    import surfel_native
    import diff_surfel_rasterization as dsr
    for _ in range(prime + warmup):
        tr.step()
    torch.cuda.synchronize()
    import gc
    ms0 = torch.cuda.memory_stats(tr.model.device)
    gc0 = [g["collections"] for g in gc.get_stats()]
    retimed = False
    for attempt in range(2):
        host = []
        reseed_views(tr)      # the same views as `python bench.py --workload <this leg>` times (bench.py)
        surfel_native.collect_stage_times()
        tr.pipe.debug = 3       # events around the dominant kernel (blend_bwd) only, resolved after the window
        t0 = time.perf_counter()
        for _ in range(steps):
            h0 = time.perf_counter()
            tr.step()
            host.append(time.perf_counter() - h0)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        tr.pipe.debug = 0
        dom = surfel_native.collect_stage_times()
        # a single host stall (tens of ms: seen on boxes of the pool with unchanged kernel times and no allocation, GC run or overflow
        # inside the window) must not be reported as the leg's step time: the window is timed ONCE more and the repeat is flagged
        if attempt == 0 and max(host) > 20.0 * sorted(host)[len(host) // 2] and max(host) > 0.3 * dt:
            retimed = True
            continue
        break
    ms1 = torch.cuda.memory_stats(tr.model.device)
    # hipMalloc / hipFree inside the timed window (the caching allocator missing: views of a capture differ in instance count, so
    # buffer sizes change from step to step) and the host's own time per step (enqueue only; the device runs behind)
    alloc = {"device_allocs": int(ms1.get("num_device_alloc", 0) - ms0.get("num_device_alloc", 0)),
             "device_frees": int(ms1.get("num_device_free", 0) - ms0.get("num_device_free", 0)),
             "alloc_retries": int(ms1.get("num_alloc_retries", 0) - ms0.get("num_alloc_retries", 0)),
             "host_ms_per_step_median": round(sorted(host)[len(host) // 2] * 1e3, 4), "host_ms_per_step_max": round(max(host) * 1e3, 4),
             "python_gc_collections_gen012": [g["collections"] - c for g, c in zip(gc.get_stats(), gc0)],
             "window_retimed_after_host_stall": retimed, "capacity_table_evictions": int(surfel_native.load().surfel_debug_capacity_evictions())}
    cam = tr.cams[0]
    W, H = int(cam.image_width), int(cam.image_height)
    st, cen = window_census(tr, steps, W, H)
    st.update(dom)      # the dominant kernel's duration is the one measured INSIDE the timed window
    # both blend_bwd walks on this very state (the library default picks one per frame on the device): data for that choice
    ab, lanes = {}, {}
    lib = surfel_native.load()
    for name, v in ((("rows", 0), ("quad", 1), ("scan", 3)) if walks else ()):
        lib.surfel_set_option(b"bwd_variant", v)
        tr.pipe.debug = 2
        # every walk's window sees the SAME views in the same order (the views of a capture differ 2-3x in instance count: windows
        # over different random views are not comparable)
        for _ in range(3):      # (cold code: the first launches of a walk that has not run yet are 25 % slower)
            tr.step()
        torch.cuda.synchronize()
        surfel_native.collect_stage_times()
        if hasattr(tr, "_rng") and hasattr(tr, "_stack"):
            tr._rng.seed(90210); tr._stack = []
        for _ in range(8):
            tr.step()
        torch.cuda.synchronize()
        t = surfel_native.collect_stage_times()
        ab[name] = round(t["blend_bwd"][0] / t["blend_bwd"][1], 4)
        # instrumented twin of the walk: lane slots issued vs lanes that held a composited (pixel, surfel) pair (one step)
        bs = torch.zeros(8, dtype=torch.int64, device=tr.model.device)
        lib.surfel_debug_set_blend_stats(surfel_native.ptr(bs))
        tr.pipe.debug = 0
        try:
            tr.step()
            torch.cuda.synchronize()
        finally:
            lib.surfel_debug_set_blend_stats(None)
        sv = bs.cpu().numpy()
        lanes[name] = {"useful_lane_frac": round(float(sv[1]) / max(1.0, float(sv[0])), 4), "wave_visits": int(sv[2]), "useful_pairs": int(sv[1])}
    lib.surfel_set_option(b"bwd_variant", 2)
    tr.pipe.debug = 0
    tiles = ((W + 15) // 16) * ((H + 15) // 16)
    R, Rs, V = cen["R"], cen["Rs"], cen["V"]
    from bench import roofline_object
    n_pass = -(-(32 + max(1, (tiles - 1).bit_length())) // 8)
    roof = roofline_object({k: v[0] / v[1] for k, v in st.items()}, workload or "?", int(tr.model.P), V, R, Rs, W, H, n_pass)
    if roof:
        roof["window"] = cen
        try:
            add_insts_per_pair(roof, useful_pairs(tr))
        except Exception:      # noqa: BLE001 — an extra
            pass
    return {"roofline": roof, "instances_staged": round(Rs, 1), "ms_per_step": round(dt / steps * 1e3, 4), "iters_per_s": round(steps / dt, 2), "steps": steps, "P": int(tr.model.P),
            "visible": round(V, 1), "instances_R": round(R, 1), "inst_per_tile": round(R / tiles, 1),
            "inst_per_surfel": round(R / max(1, int(tr.model.P)), 2), "loss": round(float(tr.last["loss"]), 5),
            "kernels_ms": {k: round(v[0] / v[1], 4) for k, v in st.items()}, "blend_bwd_ms_by_walk": ab, "blend_bwd_lanes_by_walk": lanes,
            "lazy_overflows": int(getattr(tr, "lazy_overflows", 0)), "timed_window": alloc}


def config_leg(dev, workload, steps=20, warmup=5, prime=15, walks=True):
    """One more synthetic configuration through the same full training iteration (C4 = BASELINE configs[3] per-GPU shape,
    C2H = heavy footprints)."""
    import torch
    import synthetic
    import diff_surfel_rasterization as dsr
    P, W, H, zf = synthetic.CONFIGS[workload]
    tr = make_trainer(dev, workload, n_views=8)
    out = time_trainer(tr, steps, warmup, prime=prime, workload=workload, walks=walks)
    out["workload"] = ("%s-synthetic: %d random surfels, %dx%d, median 1-sigma radius %.1f px, full iteration as the headline leg"
                       % (workload, P, W, H, synthetic.PX_RADIUS.get(workload) or max(4.0 * W / 1920.0, 1.5)))
    del tr
    dsr.set_grad_arena(None)
    torch.cuda.empty_cache()
    return out


# (the knobs of the garden preset can be overridden from the environment: that is how it was tuned to settle above a million surfels)
GARDEN_GT = int(os.environ.get("GARDEN_GT", 4_000_000)); GARDEN_INIT = int(os.environ.get("GARDEN_INIT", 5_400_000))
GARDEN_VIEWS = int(os.environ.get("GARDEN_VIEWS", 32)); GARDEN_ITERS = int(os.environ.get("GARDEN_ITERS", 2500)); GARDEN_PX = float(os.environ.get("GARDEN_PX", 0.010))
TRAINED_PRESETS = {
    # name: ground-truth surfels, initial random points, views, (W, H), iterations of the reference schedule (untimed), gt disc scale
    "trained": dict(n_gt=200_000, n_init=200_000, n_views=48, res=(800, 800), train_iters=6000, px_scale=0.035, init="cube"),
    # BASELINE configs[3]'s per-GPU shape (Mip-NeRF360 garden: ~2 M surfels, 1600x1060) on post-densification statistics: the capture
    # is dense enough for the densification to settle above a million surfels
    # is dense enough, and the initial points lie near its surface (as SfM points do), for the densification to settle above a million
    "garden": dict(n_gt=GARDEN_GT, n_init=GARDEN_INIT, n_views=GARDEN_VIEWS, res=(1600, 1060), train_iters=GARDEN_ITERS, px_scale=GARDEN_PX, init="surface"),
}


def trained_state(dev, preset="trained", state=None):
    """(model, train cameras, held-out cameras, extent, info) of a TRAINED state: the reference's default schedule (random-point
    initialisation, densification 500 -> every 100, opacity resets, lambda_dist after 3000) run — untimed — on a synthetic capture.
    `state`: a .ply path — loaded if it exists (the capture is regenerated, it is deterministic), written otherwise, so that several
    processes (the rocprofv3 passes of scripts/profile_gpu.sh) time the same model without training it again."""
    import torch
    import surfel_model
    import surfel_trainer as TR
    c = TRAINED_PRESETS[preset]
    W, H = c["res"]
    torch.manual_seed(0)
    bg = torch.zeros(3, device=dev)
    gt = TR.synthetic_object(c["n_gt"], dev, seed=0, px_scale=c["px_scale"])
    cams = TR.capture_views(gt, TR.orbit_cameras(c["n_views"] + 8, W, H, device=dev), bg)
    gt_xyz = gt._xyz.detach().cpu().numpy() if c["init"] == "surface" else None
    del gt
    train_cams, test_cams = cams[:c["n_views"]], cams[c["n_views"]:]
    extent = TR.cameras_extent(train_cams)
    model = surfel_model.GaussianModel(3, device=dev)
    info = {"preset": preset}
    if state and os.path.exists(state):
        model.load_ply(state)
        model.active_sh_degree = 3
        model.spatial_lr_scale = extent
        info["loaded_from"] = os.path.basename(state)
    else:
        rng = np.random.default_rng(0)
        pcd = type("PCD", (), {})()
        if gt_xyz is not None:      # points near the captured surface, 2 % of the object's radius off
            pick = rng.integers(0, gt_xyz.shape[0], c["n_init"])
            pcd.points = (gt_xyz[pick] + 0.024 * rng.standard_normal((c["n_init"], 3))).astype(np.float32)
        else:
            pcd.points = (rng.random((c["n_init"], 3)) * 2.6 - 1.3).astype(np.float32)
        pcd.colors = rng.random((c["n_init"], 3)).astype(np.float32)
        model.create_from_pcd(pcd, spatial_lr_scale=extent)
        opt = TR.optimization_params(iterations=c["train_iters"], lambda_dist=100.0, position_lr_max_steps=c["train_iters"])
        # (solo: under torch.distributed every rank prepares the SAME state by itself — seeded capture and points, bit-reproducible
        # gradients —; trained_trainer checks that and falls back to rank 0's state)
        tr = TR.Trainer(model, train_cams, opt, TR.pipeline_params(depth_ratio=1.0), extent=extent, solo=True)
        p0 = tr.evaluate(train_cams[:8])[0]
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(c["train_iters"] - 1):       # the schedule's last iteration takes no optimiser step (train.py:136)
            tr.step()
        torch.cuda.synchronize(); t_train = time.perf_counter() - t0
        info.update({"train_wall_s": round(t_train, 2), "train_iters_per_s": round((c["train_iters"] - 1) / t_train, 1), "psnr_init": round(p0, 2),
                     "psnr_train": round(tr.evaluate(train_cams[:8])[0], 2), "psnr_heldout": round(tr.evaluate(test_cams)[0], 2)})
        del tr
        if state:
            model.save_ply(state)
    sc = model._av["scaling"]; op = model._av["opacity"]
    info["model_stats"] = {"median_scale": round(float(sc.median()), 5), "median_opacity": round(float(op.median()), 4),
                           "frac_opacity_gt_0.5": round(float((op > 0.5).float().mean()), 4)}
    info["workload"] = ("%s: trained synthetic capture, %d initial points -> %d surfels after %d iterations of the reference schedule, %d views of %dx%d"
                        % (preset, c["n_init"], int(model.P), c["train_iters"], c["n_views"], W, H))
    return model, train_cams, test_cams, extent, info


def _adopt_rank0_state(model, extent):
    """N > 1: if the ranks' independently prepared states differ in size, every rank takes rank 0's parameters (fresh optimiser state)."""
    import torch
    import torch.distributed as dist
    import surfel_model
    if not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
        return model
    dev = model.device
    p = torch.tensor([model.P], dtype=torch.int64, device=dev)
    lo, hi = p.clone(), p.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN); dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    if int(lo) == int(hi):
        return model
    dist.broadcast(p, src=0)
    if dist.get_rank() != 0:
        fresh = surfel_model.GaussianModel(3, device=dev)
        fresh._alloc(int(p))
        fresh.max_radii2D = torch.zeros((int(p),), device=dev)
        fresh.active_sh_degree, fresh.spatial_lr_scale = 3, extent
        model = fresh
    else:
        model.grad = model.m = model.v = None      # every rank starts the timed Trainer with fresh moments
    dist.broadcast(model.theta, src=0)
    model.refresh_activations()
    return model


def trained_trainer(dev, preset="trained", state=None):
    """A Trainer on the steady-state iteration (every loss term on, no densification inside the timed window) of a trained state."""
    import surfel_trainer as TR
    model, train_cams, test_cams, extent, info = trained_state(dev, preset, state)
    model = _adopt_rank0_state(model, extent)
    tr = TR.Trainer(model, train_cams, _frozen_schedule(TR), TR.pipeline_params(depth_ratio=1.0), extent=extent)
    tr.iteration = TRAINED_PRESETS[preset]["train_iters"]
    return tr, info


def trained_leg(dev, preset="trained", steps=30, warmup=5, state=None):
    """A TRAINED state instead of random surfels (post-densification scale / opacity statistics): the steady-state iteration on it."""
    import torch
    import diff_surfel_rasterization as dsr
    tr, info = trained_trainer(dev, preset, state)
    out = time_trainer(tr, steps, warmup, prime=max(5, len(tr.cams)), workload=preset)      # (an epoch of the view stack: every view's buffer sizes seen before the window)
    out.update(info)
    model = tr.model
    del tr, model
    dsr.set_grad_arena(None)
    torch.cuda.empty_cache()
    return out


# Reference box of `ms_per_step_normalised`: the probe values of ONE box of the pool (round 5's calibration run on the final probe code,
# profiles/r05_box_probe_calibration.json: C2 step 0.6032 ms there) and the split of the C2 step by what bounds its kernels
# (profiles/r05_C2_roofline.md: blend fwd + bwd = issue-bound, binning + loss + small launches = latency-bound, Adam + preprocess
# fwd / bwd = HBM-bound).  The blend share is scaled by the frozen blend-like mix kernel, not by the pure FMA stream: boxes that differ
# by 30 % on v_fma_f32 (526 ... 779 G wave-inst/s measured in one afternoon) differ by 4 % on the step.
BOX_REF = {"blend_mix_Mvisits_per_s": 3857.0, "sort_512k_us": 53.4, "hbm_copy_GBps": 5319.0, "launch_us": 2.45, "valu_Ginst_per_s": 735.0}
STEP_SPLIT = {"blend": 0.48, "latency": 0.27, "hbm": 0.25}


def box_probe(dev):
    """What THIS box sustains, measured before the timed window: HBM copy GB/s, the cost of a dependent launch boundary,
    the wave-instruction rate and shader clock of independent v_fma_f32 streams (include/surfel_hip.h: surfel_debug_box_probe), and a
    fixed 0.5 M-pair tile sort (two look-back passes of the product's own sort on 12-bit keys: the latency-bound stage that stretched
    26 - 29 % on round 4's driver box), a frozen blend-like instruction-mix kernel.  `slowdown_vs_reference_box` = the factor by which a C2
    step is expected to be longer on this box than on the reference box, from the three classes of kernels the step consists of
    (STEP_SPLIT); a first-order model — the raw probe values are printed so that a reader can weigh them differently.  (It came out at
    1.00 - 1.08 on every box sampled in round 5 while the C2 step ranged 0.576 - 0.644 ms: the pool's run-to-run difference is not in
    anything these probes measure; bench.py normalises by the in-step tile sort instead and says so.)"""
    import ctypes as C
    import torch
    import surfel_native as n
    lib = n.load()
    scratch = torch.zeros(1 << 17, dtype=torch.uint8, device=dev)
    out = (C.c_float * 12)()
    s = n.current_stream_ptr(dev)
    res = {}
    with torch.cuda.device(dev):
        for _ in range(2):      # (first call: code upload)
            rc = lib.surfel_debug_box_probe(n.ptr(scratch), scratch.numel(), out, s)
        if rc != 0:
            return {"error": n.last_error()}
        res.update({"launch_us": round(out[0], 3), "valu_Ginst_per_s": round(out[1], 1), "valu_Ginst_per_s_in_kernel_span": round(out[5], 1),
                    "shader_clock_GHz_under_fma_grid": round(out[2], 3), "fma_cycles_per_wave_inst_per_simd": round(out[4], 3),
                    "add_cycles_per_wave_inst_per_simd": round(out[6], 3), "pk_fma_cycles_per_wave_inst_per_simd": round(out[7], 3),
                    "blend_mix_Mvisits_per_s": round(out[8], 1),
                    "note": "box-relative figures: 1024 workgroups of 4 waves, 8 FMAs + 3 scalar loop instructions per trip, span timing.  The instruction classes "
                            "themselves are measured by scripts/issue_probe.hip (profiles/r05_issue_probe.md): 2.5 shader cycles per wave64 instruction per SIMD for "
                            "fma / add / mul / mov / and, 4.3 for DPP / min / max / compare / select / shift / SGPR-operand forms, 8.2 transcendental; two waves per SIMD "
                            "reach those rates: the non-packed fp32 ceiling is 1024 SIMDs x clock / 2.5"})
        # the fixed sort: 512 Ki (key, value) pairs, 12-bit keys (an 800x800 frame's tile ids), the product's own passes
        N = 1 << 19
        g = torch.Generator(device="cpu").manual_seed(1)
        keys0 = torch.randint(0, 2500, (N,), generator=g, dtype=torch.int32).to(dev)
        vals0 = torch.arange(N, dtype=torch.int32, device=dev)
        alloc = n.TorchAllocator(dev)
        times = []
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for k in range(12):
            keys, vals = keys0.clone(), vals0.clone()
            e0.record()
            rc = lib.surfel_debug_sort_pairs(alloc.cb, None, n.ptr(keys), n.ptr(vals), N, 0, 12, s)
            e1.record()
            torch.cuda.synchronize()
            if rc != 0:
                return {"error": n.last_error()}
            if k >= 2:
                times.append(e0.elapsed_time(e1) * 1e3)
        res["sort_512k_us"] = round(float(np.median(times)), 2)
        assert bool((keys[1:] >= keys[:-1]).all())
    cp = copy_bandwidth(dev, mbytes=512, iters=8)
    res["hbm_copy_GBps"] = cp["GBps_read_plus_write"]
    # dependent-load latency (one lane chasing a cycle): L2-sized and HBM-sized footprints
    with torch.cuda.device(dev):
        lat = (C.c_float * 1)()
        for name, mb, hops in (("latency_ns_4MiB", 4, 20000), ("latency_ns_1GiB", 1024, 20000)):
            buf = torch.empty(mb << 20, dtype=torch.uint8, device=dev)
            res[name] = round(lat[0], 1) if lib.surfel_debug_latency_probe(n.ptr(buf), buf.numel(), hops, lat, s) == 0 else None
            del buf
        torch.cuda.empty_cache()
    # the HOST: what enqueueing costs on this box's CPU — a tiny torch kernel launch (asynchronous) and a ctypes round trip
    x = torch.zeros(64, device=dev)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(2000):
        x.add_(1.0)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t1b = time.perf_counter()
    for _ in range(5000):
        lib.surfel_abi_version()
    t2 = time.perf_counter()
    acc = 0
    for k in range(200000):
        acc += k & 3
    t3 = time.perf_counter()
    res["host_torch_launch_us"] = round((t1 - t0) / 2000 * 1e6, 2)
    res["host_ctypes_call_us"] = round((t2 - t1b) / 5000 * 1e6, 3)
    res["host_python_loop_ns_per_iter"] = round((t3 - t2) / 200000 * 1e9, 1)
    r = BOX_REF
    res["reference_box"] = dict(r)
    res["slowdown_vs_reference_box_by_probes"] = round(STEP_SPLIT["blend"] * r["blend_mix_Mvisits_per_s"] / max(res["blend_mix_Mvisits_per_s"], 1e-3)
                                             + STEP_SPLIT["latency"] * res["sort_512k_us"] / r["sort_512k_us"]
                                             + STEP_SPLIT["hbm"] * r["hbm_copy_GBps"] / max(res["hbm_copy_GBps"], 1e-3), 4)
    res["step_split_assumed"] = dict(STEP_SPLIT)
    return res


def copy_bandwidth(dev, mbytes=1024, iters=10):
    """Same-run HBM copy probe: device-to-device copy of a buffer far larger than the 256 MB Infinity Cache;
    GB/s counts bytes read + written."""
    import torch
    n = mbytes * 1024 * 1024 // 4
    a = torch.empty(n, dtype=torch.float32, device=dev).normal_()
    b = torch.empty_like(a)
    for _ in range(2):
        b.copy_(a)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        b.copy_(a)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    del a, b
    torch.cuda.empty_cache()
    return {"buffer_MB": mbytes, "ms": round(ms, 4), "GBps_read_plus_write": round(2 * mbytes * 1.048576 / ms, 1)}


def cpu_dense_c1(max_seconds=25.0):
    """BASELINE configs[0]: the dense pure-PyTorch surfel rasterizer (oracle/dense_autograd.py — every surfel against every pixel,
    the algorithm of the reference's linked Python notebook, README.md:3,7) forward on the host CPU at C1 (10 k surfels, 256x256),
    on a bounded sample of pixel rows, extrapolated linearly to the full image (the work is uniform per pixel row)."""
    import torch
    import synthetic
    from oracle import dense_autograd as da
    from oracle.surfel_oracle import Oracle
    from helpers import oracle_forward, scene_args      # tests/helpers.py (sys.path is set by bench.py)
    P, W, H, zf = synthetic.CONFIGS["C1"]
    sc = synthetic.make_scene(P, W, H, seed=0, z_far=zf)
    a = scene_args(sc)
    _, _, _, radii, st = oracle_forward(Oracle("f64"), a)
    return da.time_dense_forward(sc, st, max_seconds, cores=CPU_THREADS)


def train_iter(dev, workload="C2", iters=60, warmup=15, n_views=8):
    """Full training iterations/s (one view per iteration) of make_trainer's job, timed on its own."""
    import torch
    import synthetic
    P, W, H, zf = synthetic.CONFIGS[workload]
    tr = make_trainer(dev, workload, n_views)
    l0 = None
    for i in range(warmup):
        tr.step()
        if i == 0:
            l0 = float(tr.last["loss"])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        tr.step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return {"workload": "%s-synthetic surfels (%d), %dx%d, %d views, full iteration: fwd + L1/SSIM + normal/dist regularisers + bwd + stats + Adam"
                        % (workload, P, W, H, n_views), "iters_per_s": round(iters / dt, 2), "ms_per_iter": round(dt / iters * 1e3, 4),
            "loss_first": round(l0, 5), "loss_last": round(float(tr.last["loss"]), 5), "iters": iters, "warmup": warmup}


def raster_fwd_bwd(dev, workload="C2", iters=40, warmup=10):
    """The rasterizer alone (forward + backward with random upstream gradients, no loss / optimiser): ms per view."""
    import torch
    import synthetic
    from diff_surfel_rasterization import GaussianRasterizationSettings, GaussianRasterizer
    import diff_surfel_rasterization as dsr
    dsr.set_grad_arena(None)
    if workload not in synthetic.CONFIGS:      # trained workloads: the synthetic configuration of the same shape (garden -> C4: ~2 M surfels, 1600x1060)
        workload = "C4" if workload == "garden" else "C2"
    P, W, H, zf = synthetic.CONFIGS[workload]
    sc = synthetic.make_scene(P, W, H, seed=0, z_far=zf, px_radius=synthetic.PX_RADIUS.get(workload))
    t = lambda x: torch.as_tensor(np.ascontiguousarray(x)).to(dev)
    rs = GaussianRasterizationSettings(image_height=H, image_width=W, tanfovx=sc["tanfovx"], tanfovy=sc["tanfovy"], bg=t(sc["bg"]),
                                       scale_modifier=1.0, viewmatrix=t(sc["viewmatrix"]), projmatrix=t(sc["projmatrix"]),
                                       sh_degree=3, campos=t(sc["campos"]), prefiltered=False, debug=False)
    rast = GaussianRasterizer(raster_settings=rs)
    params = [t(sc[k]).requires_grad_(True) for k in ("means3D", "shs", "opacities", "scales", "rotations")]
    g = torch.Generator(device="cpu").manual_seed(1234)
    gC = torch.randn((3, H, W), generator=g).to(dev); gO = torch.randn((7, H, W), generator=g).to(dev)

    def step():
        m2 = torch.zeros_like(params[0], requires_grad=True)
        color, radii, allmap = rast(means3D=params[0], means2D=m2, shs=params[1], colors_precomp=None, opacities=params[2], scales=params[3],
                                    rotations=params[4], cov3D_precomp=None)
        torch.autograd.backward([color, allmap], [gC, gO])
        for p_ in params:
            p_.grad = None
    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return {"workload": "%s-synthetic, rasterizer forward+backward only through the drop-in autograd module (diff_surfel_rasterization.GaussianRasterizer over the C ABI; dL/dsh written), N(0,1) upstream gradients" % workload,
            "ms_per_view": round(dt / iters * 1e3, 4), "views_per_s": round(iters / dt, 2), "Msplats_per_s": round(P * iters / dt / 1e6, 2)}


def fwd_1080p_sweep(dev, sizes=(300_000, 1_000_000, 2_000_000, 5_000_000, 10_000_000), iters=8, warmup=3):
    """forward-only throughput at 1920x1080 over P in {0.3, 1, 2, 5, 10} M random surfels (the 1 M point is
    fwd_1080p's workload): Msplats/s, M tile instances/s and the forward's HBM roofline fraction per size."""
    import synthetic
    W, H = 1920, 1080
    zf = synthetic.CONFIGS["1080p_1M"][3]
    out = []
    for P in sizes:
        med, R, Rs = _forward_only(dev, int(P), W, H, zf, iters, warmup)
        rf = fwd_roofline(int(P), R, Rs, W, H, med)
        out.append({"P": int(P), "ms_median": round(med, 4), "Msplats_per_s": round(P / (med * 1e-3) / 1e6, 1), "instances_R": R, "instances_staged": Rs,
                    "Minst_per_s": round(R / (med * 1e-3) / 1e6, 1), "hbm_frac": rf["frac"], "hbm_frac_survey_R": rf["survey_R"]["frac"]})
    return {"workload": "forward only (preprocess .. blend incl. sort) at 1920x1080 under no_grad, random surfels (synthetic.make_scene), median of %d calls" % iters, "sizes": out}


def full_train_leg(dev, iterations=30_000, res=(800, 600), n_views=49, n_gt=60_000, n_init=40_000, max_seconds=240.0):
    """BASELINE configs[2] as a FULL TRAIN: the reference's 30 000-iteration schedule (arguments/__init__.py:75-95: densification
    500 -> 15 000 every 100, opacity reset every 3 000, position-lr decay over 30 000) with the DTU settings of scripts/dtu_eval.py:23
    (depth_ratio 1, lambda_dist 1000; lambda_normal 0.05 from 7 000, distortion from 3 000: train.py:80-81) on an 800x600 synthetic
    capture of 49 views (the DTU scans' view count) initialised from random points (scene/dataset_readers.py:236-242) — the whole
    loop of train.py:54-140 including what the steady-state legs leave out: densify / clone / split / prune and the opacity resets
    (torch indexing on the parameter store, host-driven), timed separately (a device synchronisation on each side of the ~150 events).
    Reports wall seconds, iterations/s per phase, surfels over time, PSNR (train views / 8 held-out views)."""
    import torch
    import surfel_model
    import surfel_trainer as TR
    import diff_surfel_rasterization as dsr
    W, H = res
    torch.manual_seed(0)
    bg = torch.zeros(3, device=dev)
    gt = TR.synthetic_object(n_gt, dev, seed=3, px_scale=0.03)
    cams = TR.capture_views(gt, TR.orbit_cameras(n_views + 8, W, H, device=dev), bg)
    del gt
    train_cams, test_cams = cams[:n_views], cams[n_views:]
    extent = TR.cameras_extent(train_cams)
    rng = np.random.default_rng(1)
    pcd = type("PCD", (), {})()
    pcd.points = (rng.random((n_init, 3)) * 2.6 - 1.3).astype(np.float32)
    pcd.colors = rng.random((n_init, 3)).astype(np.float32)
    model = surfel_model.GaussianModel(3, device=dev)
    model.create_from_pcd(pcd, spatial_lr_scale=extent)
    opt = TR.optimization_params(iterations=iterations, lambda_dist=1000.0, lambda_normal=0.05)
    tr = TR.Trainer(model, train_cams, opt, TR.pipeline_params(depth_ratio=1.0), extent=extent)
    psnr0 = tr.evaluate(train_cams[:8])[0]
    # time spent in the schedule's events (densify_and_prune, reset_opacity): wrap them, synchronising on both sides
    ev = {"densify_prune_s": 0.0, "densify_prune_calls": 0, "opacity_reset_s": 0.0, "opacity_reset_calls": 0}

    def timed(fn, key):
        def wrapper(*a, **k):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            r = fn(*a, **k)
            torch.cuda.synchronize()
            ev[key + "_s"] += time.perf_counter() - t0; ev[key + "_calls"] += 1
            return r
        return wrapper
    model.densify_and_prune = timed(model.densify_and_prune, "densify_prune")
    model.reset_opacity = timed(model.reset_opacity, "opacity_reset")
    phases = [("warm-up (1 - 500)", 500), ("densification (501 - 15 000)", opt.densify_until_iter), ("refinement (15 001 - 30 000)", iterations)]
    points, phase_out = [[0, int(model.P)]], []
    torch.cuda.synchronize()
    t_start = time.perf_counter()
    it, cut = 0, False
    for name, until in phases:
        t0, it0, ev0 = time.perf_counter(), it, dict(ev)
        while it < min(until, iterations):
            tr.step(); it += 1
            if it % 1000 == 0:
                points.append([it, int(model.P)])
                if time.perf_counter() - t_start > max_seconds:      # (bounded: a slow box must not cost the bench its other legs)
                    cut = True
                    break
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        phase_out.append({"phase": name, "iterations": it - it0, "wall_s": round(dt, 2), "iters_per_s": round((it - it0) / max(dt, 1e-9), 1),
                          "of_which_densify_prune_s": round(ev["densify_prune_s"] - ev0["densify_prune_s"], 3),
                          "of_which_opacity_reset_s": round(ev["opacity_reset_s"] - ev0["opacity_reset_s"], 3), "points_at_end": int(model.P)})
        if cut:
            break
    wall = time.perf_counter() - t_start
    out = {"workload": "config-3 full train: %d iterations of the reference schedule on a synthetic %dx%d capture (%d train views, %d random initial points, "
                       "ground truth = %d surfels), depth_ratio 1, lambda_dist 1000 from 3 000, lambda_normal 0.05 from 7 000, densify 500 -> 15 000 / 100, "
                       "opacity reset / 3 000" % (it, W, H, n_views, n_init, n_gt),
           "iterations_done": it, "completed": not cut, "wall_s": round(wall, 2), "iters_per_s": round(it / wall, 1), "phases": phase_out,
           "events": {k: (round(v, 3) if isinstance(v, float) else v) for k, v in ev.items()},
           "points_over_time": points, "points_final": int(model.P), "lazy_overflows": int(tr.lazy_overflows),
           "psnr_init": round(psnr0, 2), "psnr_train": round(tr.evaluate(train_cams[:8])[0], 2), "psnr_heldout": round(tr.evaluate(test_cams)[0], 2)}
    del tr, model
    dsr.set_grad_arena(None)
    torch.cuda.empty_cache()
    return out

