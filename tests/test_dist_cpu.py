"""CPU tests of the multi-GPU shim: world_size-2 gloo processes for the collectives; the tile-band projection trick
is checked against the fp64 oracle (band images concatenate to the full image, band gradients sum to the full gradient)."""
import os
import socket
from collections import namedtuple

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _fake_grads(P, rank):
    g = torch.Generator().manual_seed(100 + rank)
    shapes = dict(xyz=(P, 3), f_dc=(P, 1, 3), f_rest=(P, 15, 3), opacity=(P, 1), scaling=(P, 2), rotation=(P, 4))
    return {k: torch.randn(s, generator=g) for k, s in shapes.items()}


def _worker(rank, world, port, P, q):
    import sys
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(here, "2d-gaussian-splatting_amd"))
    import surfel_dist as sd
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        grads = _fake_grads(P, rank)
        b = sd.GradBucket(P, "cpu")
        b.pack(grads); b.all_reduce(average=False)
        out = b.unpack(grads)
        expect = {k: sum(_fake_grads(P, r)[k] for r in range(world)) for k in grads}
        ok = all(torch.allclose(out[k], expect[k], atol=1e-6) for k in grads)
        b.pack(grads); b.all_reduce(average=True)
        ok &= torch.allclose(b.unpack(grads)["xyz"], expect["xyz"] / world, atol=1e-6)
        # densification statistics
        g = torch.Generator().manual_seed(7 + rank)
        gn = torch.rand((P, 1), generator=g); vis = torch.rand(P, generator=g) > 0.5; rad = torch.randint(0, 30, (P,), generator=g)
        sn, dn, mr = sd.reduce_densification_stats(gn, vis, rad)
        e_sn = torch.zeros(P, 1); e_dn = torch.zeros(P, 1); e_mr = torch.zeros(P)
        for r in range(world):
            g2 = torch.Generator().manual_seed(7 + r)
            gn2 = torch.rand((P, 1), generator=g2); vis2 = torch.rand(P, generator=g2) > 0.5; rad2 = torch.randint(0, 30, (P,), generator=g2)
            e_sn += gn2 * vis2[:, None]; e_dn += vis2[:, None].float(); e_mr = torch.maximum(e_mr, torch.where(vis2, rad2.float(), torch.zeros(())))
        ok &= torch.allclose(sn, e_sn, atol=1e-6) and torch.equal(dn, e_dn) and torch.equal(mr, e_mr)
        # parameter broadcast
        p = torch.full((4, 3), float(rank))
        sd.broadcast_parameters([p], src=0)
        ok &= bool((p == 0).all())
        # view schedule: the ranks of one step get distinct views; every rank computes the same schedule
        mine = [sd.view_indices(10, world, rank, it, seed=3) for it in range(12)]
        other = [sd.view_indices(10, world, 1 - rank, it, seed=3) for it in range(12)]
        ok &= all(a != b_ for a, b_ in zip(mine, other))
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


def test_gloo_world2_collectives():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, 257, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    assert res == [(0, True), (1, True)]


def test_band_bounds_never_shorter_than_the_halo():
    """H = 1080 (67.5 tile rows) with bottom-heavy weights used to end in a (1056, 1080) band of 24 rows — shorter than the
    32-row halo the band-sharded loss exchanges, so the step after a re-balance raised.  A trailing partial granule now belongs to
    the granule before it."""
    import surfel_dist as sd
    for H in (1080, 1060, 2160, 600, 97):
        rows16 = (H + 15) // 16
        for world in (1, 2, 3, 4, 8):
            for w in (None, [1.0] * (rows16 - 1) + [1e6], [1e6] + [1.0] * (rows16 - 1), [float(i * i) for i in range(rows16)]):
                b = sd.band_bounds(H, world, w, multiple=sd.HALO)
                assert b[0][0] == 0 and b[-1][1] == H and all(b[i][1] == b[i + 1][0] for i in range(world - 1)), (H, world, b)
                live = [y1 - y0 for y0, y1 in b if y1 > y0]
                if H // sd.HALO >= world:
                    assert len(live) == world and min(live) >= sd.HALO, (H, world, w is None, b)
                assert all(y0 % sd.HALO == 0 for y0, _ in b if y0 < H), b


def test_band_bounds():
    import surfel_dist as sd
    b = sd.band_bounds(2160, 8)
    assert b[0][0] == 0 and b[-1][1] == 2160 and all(b[i][1] == b[i + 1][0] for i in range(7)) and all(y0 % 16 == 0 for y0, _ in b)
    assert max(y1 - y0 for y0, y1 in b) - min(y1 - y0 for y0, y1 in b) <= 32
    w = [1.0] * 10 + [9.0] * 5            # heavy bottom rows -> the bottom band is shorter
    b2 = sd.band_bounds(240, 2, w)
    assert b2[0][1] > 120 and b2[1][1] == 240


def test_tile_band_sharding_matches_full_render():
    """Rendering rows [y0,y1) through band_settings' shifted projection == the same rows of the full render, and the
    per-surfel gradients of the bands add up to the full-image gradients (fp64 oracle)."""
    import surfel_dist as sd
    import synthetic
    from helpers import oracle_forward, scene_args
    from oracle.surfel_oracle import Oracle
    RS = namedtuple("RS", "image_height image_width tanfovx tanfovy bg scale_modifier viewmatrix projmatrix sh_degree campos prefiltered debug")
    W, H, P = 96, 80, 700
    sc = synthetic.make_scene(P, W, H, seed=5, px_radius=5.0)
    a = scene_args(sc)
    o = Oracle("f64")
    R, col, oth, radii, st = oracle_forward(o, a)
    rng = np.random.default_rng(1)
    gC = rng.normal(size=col.shape); gO = rng.normal(size=oth.shape)
    gO[5] = 0     # median-depth channel is piecewise constant; keep its gradient out of the additivity check
    g_full = o.rasterize_backward(st, gC, gO)
    rs = RS(H, W, a["tanfovx"], a["tanfovy"], torch.tensor(a["bg"]), 1.0, torch.tensor(a["viewmatrix"]), torch.tensor(a["projmatrix"]), 3,
            torch.tensor(a["campos"]), False, False)
    acc = None
    for (y0, y1) in sd.band_bounds(H, 2):
        rb = sd.band_settings(rs, y0, y1)
        ab = dict(a); ab["H"] = rb.image_height; ab["projmatrix"] = rb.projmatrix.numpy(); ab["tanfovy"] = rb.tanfovy
        Rb, colb, othb, radb, stb = oracle_forward(o, ab)
        assert np.abs(colb - col[:, y0:y1]).max() < 1e-5            # projection matrices are fp32: 1e-7-relative shift
        assert np.abs(othb[[0, 1, 2, 3, 4, 6]] - oth[[0, 1, 2, 3, 4, 6], y0:y1]).max() < 1e-4
        gb = o.rasterize_backward(stb, gC[:, y0:y1], gO[:, y0:y1])
        part = [gb.dL_dmeans3D, gb.dL_dscales, gb.dL_drots, gb.dL_dopacity, gb.dL_dsh]
        acc = part if acc is None else [x + y for x, y in zip(acc, part)]
    for got, ref in zip(acc, [g_full.dL_dmeans3D, g_full.dL_dscales, g_full.dL_drots, g_full.dL_dopacity, g_full.dL_dsh]):
        assert np.abs(got - ref).max() <= 2e-3 * np.abs(ref).max()


def _trainer_stats_worker(rank, world, port, q):
    """The trainer's N>1 collectives on the flat store, with gloo on CPU tensors (the kernels themselves need a GPU)."""
    import os
    import types
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import surfel_trainer as TR
        P = 11
        model = types.SimpleNamespace(
            xyz_gradient_accum=torch.arange(P, dtype=torch.float32).reshape(P, 1) * (rank + 1),
            denom=torch.full((P, 1), float(rank + 1)),
            max_radii2D=torch.arange(P, dtype=torch.float32) * (1 if rank == 0 else -1) + rank,
            grad=torch.arange(P * 58, dtype=torch.float32) * (rank + 1))
        tr = TR.Trainer.__new__(TR.Trainer)
        tr.model, tr.world, tr.rank = model, world, rank
        tr._reduce_stats()
        import surfel_model as SM
        gcol = torch.full((P, 3), float(rank + 1))
        gall = SM.exchange_collectives(model.grad, gcol, P)        # the per-iteration collectives of Trainer.step
        assert gall.shape == (world, P, 3) and all(float(gall[r].min()) == float(gall[r].max()) == r + 1 for r in range(world))
        # geometry prefix summed, SH block left to be rebuilt from the gathered colour gradients
        assert torch.equal(model.grad[10 * P:], torch.arange(P * 58, dtype=torch.float32)[10 * P:] * (rank + 1))
        model.grad[10 * P:] = torch.arange(P * 58, dtype=torch.float32)[10 * P:] * 3
        # every rank draws a different view of the same permutation
        views = [TR.surfel_dist.view_indices(9, world, rank, it, seed=3) for it in range(6)]
        # the cached per-epoch schedule the trainer uses is the same function
        per_epoch = max(1, 9 // world)
        for it in range(10):
            ep, k = divmod(it, per_epoch)
            sched = TR.surfel_dist.epoch_schedule(9, world, ep, seed=3)
            assert sched[k] == [TR.surfel_dist.view_indices(9, world, r, it, seed=3) for r in range(world)]
        q.put((rank, model.xyz_gradient_accum.reshape(-1).tolist(), model.denom.reshape(-1).tolist(), model.max_radii2D.tolist(),
               float(model.grad.sum()), views))
    finally:
        dist.destroy_process_group()


def test_trainer_collectives_world2():
    import torch
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_trainer_stats_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    P = 11
    for rank, acc, den, mr, gsum, views in res:
        assert acc == [3.0 * i for i in range(P)]                  # (1 + 2) * i
        assert den == [3.0] * P
        assert mr == [max(float(i), float(-i + 1)) for i in range(P)]
        assert gsum == 3.0 * sum(range(P * 58))
    assert all(a != b for a, b in zip(res[0][5], res[1][5]))       # distinct views per step across ranks


def _lazy_redo_worker(rank, world, port, q):
    """Whole Trainer.step()s over gloo with the device pieces faked: rank 1 is told once that its lazily counted frame overflowed."""
    import os
    import sys
    import types
    import torch
    import torch.distributed as dist
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(here, "2d-gaussian-splatting_amd"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import diff_surfel_rasterization as dsr
        import surfel_native as n
        import surfel_trainer as TR
        P = 7
        log, calls, finishes = [], [], []

        class FakeModel:
            device = torch.device("cpu")
            step_count = 0
            active_sh_degree = 0
            theta = m = v = None

            def __init__(self):
                self.P = P
                self.grad = torch.arange(P * 58, dtype=torch.float32) * (rank + 1)
                self.gcol = torch.full((P, 3), float(rank + 1))
                self.xyz_gradient_accum, self.denom, self.max_radii2D = torch.zeros(P, 1), torch.zeros(P, 1), torch.zeros(P)

            def update_learning_rate(self, it): pass
            def oneupSHdegree(self): pass
            def bind(self, sh_grad=True): pass
            def refresh_activations(self): pass
            def training_setup(self, opt): pass
            def add_densification_stats(self, g, radii=None): log.append("stats")
            def optimizer_step(self, grad_scale=1.0, colour_grads=None, parts=3): log.append(("adam", grad_scale, tuple(colour_grads[1].shape)))

        def fake_rasterize(cam, m, pipe, bg, zero_means2D=True, debug_bits=0):
            calls.append((cam.uid, debug_bits))
            return (torch.zeros(3, 4, 4, requires_grad=True), torch.ones(m.P, dtype=torch.int32), torch.zeros(7, 4, 4, requires_grad=True),
                    torch.zeros(m.P, 3, requires_grad=True))

        def fake_finish():
            finishes.append(len(calls))
            if rank == 1 and len(finishes) == 2:
                raise n.CapacityOverflow("forced")
            return 1
        TR.rasterize = fake_rasterize
        TR.train_loss = lambda image, *a, **k: (image.sum() * 0.0, torch.zeros(6))
        dsr.finish_count = fake_finish
        cams = [types.SimpleNamespace(uid=i, original_image=torch.zeros(3, 4, 4), camera_center=torch.full((3,), float(i)), post_consts=lambda: torch.zeros(24))
                for i in range(4)]
        opt = TR.optimization_params(iterations=50, densify_from_iter=10 ** 6, dist_from_iter=0, normal_from_iter=0, lambda_dist=1.0)
        tr = TR.Trainer(FakeModel(), cams, opt, TR.pipeline_params(depth_ratio=1.0), extent=1.0)
        tr.manual_chain = False
        assert tr._exchange and not tr._async_exchange and tr.lazy_count
        for _ in range(4):
            tr.step()
        q.put((rank, calls, [e for e in log if e != "stats"], log.count("stats"), tr.lazy_overflows, float(tr.model.grad[:10 * P].sum())))
    finally:
        dist.destroy_process_group()


def test_lazy_redo_on_one_rank_keeps_the_collectives_in_step():
    """View-parallel training with lazily counted forwards (Trainer.lazy_count): a rank whose frame overflowed renders, evaluates and
    back-propagates again BEFORE the iteration's collectives — the other rank simply waits there; every rank still issues one
    exchange, one statistics update and one optimiser step per iteration (two gloo processes, device pieces faked)."""
    import torch.multiprocessing as mp
    import surfel_native as n
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_lazy_redo_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    P = 7
    for rank, calls, adams, nstats, overflows, gsum in res:
        bits = [b for _, b in calls]
        if rank == 0:
            assert bits == [n.OPT_LAZY_COUNT] * 4 and overflows == 0
        else:
            assert bits == [n.OPT_LAZY_COUNT, n.OPT_LAZY_COUNT, n.OPT_EXACT_BINNING, n.OPT_LAZY_COUNT, n.OPT_LAZY_COUNT] and overflows == 1
            assert calls[1][0] == calls[2][0]                      # the same view again
        assert len(adams) == 4 and nstats == 4
        assert all(a == ("adam", 0.5, (2, P, 3)) for a in adams)   # averaged over the two views, both ranks' colour gradients gathered
    assert res[0][5] == res[1][5]                                  # the geometry prefix was all-reduced the same number of times on both ranks
    views0, views1 = [c[0] for c in res[0][1]], [c[0] for c in res[1][1] if c[1] != n.OPT_EXACT_BINNING]
    assert all(a != b for a, b in zip(views0, views1))             # distinct views per iteration across the ranks


def _gather_bands_worker(rank, world, port, q):
    import os
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import surfel_dist as sd
        H, W = 80, 6
        bounds = sd.band_bounds(H, world)                       # 80 rows = 5 tile rows -> two unequal bands
        y0, y1 = bounds[rank]
        full_ref = torch.arange(2 * H * W, dtype=torch.float32).reshape(2, H, W)
        band = full_ref[:, y0:y1].clone().requires_grad_(True)
        full = sd.gather_bands(band, bounds)
        wgt = torch.linspace(-1, 1, 2 * H * W).reshape(2, H, W)
        (full * wgt).sum().backward()
        q.put((rank, bool(torch.equal(full, full_ref)), bool(torch.equal(band.grad, wgt[:, y0:y1])), bounds))
    finally:
        dist.destroy_process_group()


def test_gather_bands_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_gather_bands_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ok_full, ok_grad, bounds in res:
        assert ok_full and ok_grad
        assert bounds[0][0] == 0 and bounds[-1][1] == 80 and bounds[0][1] == bounds[1][0] and bounds[0][1] % 16 == 0
        assert bounds[0][1] - bounds[0][0] != bounds[1][1] - bounds[1][0]          # unequal bands: exercises the padding


def _halo_worker(rank, world, port, q):
    import sys
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(here, "2d-gaussian-splatting_amd"))
    import surfel_dist as sd
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        H, W, C = 200, 24, 10
        full = torch.arange(C * H * W, dtype=torch.float32).reshape(C, H, W)
        bounds = sd.band_bounds(H, world, [1.0, 1.0, 3.0, 3.0, 1.0, 1.0, 1.0, 1.0, 2.0, 2.0, 1.0, 1.0, 1.0], multiple=sd.HALO)
        y0, y1 = bounds[rank]
        band = full[:, y0:y1].clone().requires_grad_(True)
        ext = sd.exchange_halo(band, bounds, H)
        top, bot = sd.halo_rows(bounds, rank, H)
        ok = torch.equal(ext.detach(), full[:, y0 - top:y1 + bot])                 # band + the neighbours' rows, in place
        g = torch.randn(ext.shape, generator=torch.Generator().manual_seed(rank))
        ext.backward(g)
        ok &= torch.equal(band.grad, g[:, top:top + y1 - y0])                      # halo gradients are dropped, own rows pass through
        ok &= sd.halo_bytes(bounds, rank, H, W, C) == 4 * C * W * (top + bot)
        q.put((rank, bool(ok), bounds))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_halo_exchange(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_halo_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ok, bounds in res:
        assert ok
        assert all(y0 % 32 == 0 for y0, _ in bounds) and bounds[0][0] == 0 and bounds[-1][1] == 200
        assert all(bounds[i][1] == bounds[i + 1][0] for i in range(world - 1)) and all(y1 - y0 >= 32 for y0, y1 in bounds)


def test_wire_bytes_accounting():
    import sys
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(here, "2d-gaussian-splatting_amd"))
    import surfel_dist as sd
    P = 1_000_000
    v = sd.wire_bytes_per_step(P, 8, "views", True)
    assert v["all_gather_colour"] == 7 * 12 * P and v["all_reduce_geometry"] == int(1.75 * 40 * P) and abs(v["per_surfel_B"] - 154.0) < 0.1
    b = sd.wire_bytes_per_step(P, 8, "bands", False, halo_b=12345)
    assert b["all_reduce"] == int(1.75 * 52 * P) and b["halo_p2p"] == 12345 and b["radii_max_all_reduce"] == 0
    assert sd.wire_bytes_per_step(P, 2, "bands", True)["all_reduce_per_surfel_B"] == 64
    assert sd.wire_bytes_per_step(P, 1, "views", True)["total"] == 0
