"""CPU: the control flow of surfel_trainer.Trainer.step against the reference's loop (train.py:54-138) with the kernels mocked
out — learning-rate / SH-degree schedule, lambda schedules, densification schedule, "no optimiser update on iterations that
re-create the parameters", opacity reset, view sampling without replacement."""
import types

import pytest
import torch


class FakeModel:
    def __init__(self):
        self.device = torch.device("cpu")
        self.P = 5
        self.grad = torch.zeros(5 * 58)
        self._gv = {"opacity": torch.ones(5, 1)}
        self.gcol = torch.zeros(5, 3)
        self.active_sh_degree = 0
        self.log = []

    def update_learning_rate(self, it): self.log.append(("lr", it))
    def oneupSHdegree(self): self.active_sh_degree += 1; self.log.append(("sh",))
    def bind(self, sh_grad=True): self.log.append(("bind", sh_grad))
    def add_densification_stats(self, g, radii=None): self.log.append(("stats",))
    def densify_and_prune(self, *a, **k): self.log.append(("densify", a, k.get("generator")))
    def reset_opacity(self): self.log.append(("reset",))
    def optimizer_step(self, grad_scale=1.0, colour_grads=None, parts=3): self.log.append(("adam", grad_scale, colour_grads is not None, parts))
    def training_setup(self, opt): pass

    def update_step(self, colour_grads, stats=None, grad_scale=1.0):      # statistics + Adam in one launch (surfel_train_update)
        if stats is not None:
            self.log.append(("stats",))
        self.log.append(("adam", grad_scale, True, 3))
        self.log.append(("one_launch",))


@pytest.fixture()
def trainer(monkeypatch):
    import surfel_trainer as TR
    calls = []

    def fake_rasterize(cam, m, pipe, bg, zero_means2D=True, debug_bits=0):
        calls.append(("raster", cam.uid, debug_bits))
        m2 = torch.zeros(m.P, 3, requires_grad=True)
        img = torch.zeros(3, 4, 4, requires_grad=True)
        return img, torch.ones(m.P, dtype=torch.int32), torch.zeros(7, 4, 4, requires_grad=True), m2

    def fake_train_loss(image, allmap, gt, cam, ratio, l_dssim, l_n, l_d, defer_scalars=False):
        calls.append(("loss", allmap is not None, cam is not None, l_n, l_d))
        return image.sum() * 0.0, torch.zeros(6)

    import diff_surfel_rasterization as dsr
    counts = []

    def fake_finish_count():
        counts.append(len(calls))
        if fake_finish_count.overflow_at and len(counts) in fake_finish_count.overflow_at:
            import surfel_native
            raise surfel_native.CapacityOverflow("forced")
        return 1
    fake_finish_count.overflow_at = ()
    monkeypatch.setattr(TR, "rasterize", fake_rasterize)
    monkeypatch.setattr(TR, "train_loss", fake_train_loss)
    monkeypatch.setattr(dsr, "finish_count", fake_finish_count)
    cams = [types.SimpleNamespace(uid=i, original_image=torch.zeros(3, 4, 4), camera_center=torch.zeros(3), post_consts=lambda: torch.zeros(24))
            for i in range(4)]
    m = FakeModel()
    opt = TR.optimization_params(iterations=60, densify_from_iter=10, densification_interval=10, densify_until_iter=45, opacity_reset_interval=30,
                                 dist_from_iter=5, normal_from_iter=20, lambda_dist=100.0, lambda_normal=0.05)
    tr = TR.Trainer(m, cams, opt, TR.pipeline_params(depth_ratio=1.0), extent=3.0)
    tr._fake_finish_count = fake_finish_count
    tr.manual_chain = False      # (the faked pieces are the autograd path's; the hand-driven chain is compared with it on the GPU)
    return tr, m, calls, opt


def test_loop_schedules_match_reference(trainer):
    tr, m, calls, opt = trainer
    per_it = []
    for _ in range(opt.iterations):
        n0, c0 = len(m.log), len(calls)
        tr.step()
        per_it.append((m.log[n0:], calls[c0:]))
    for it, (log, cl) in enumerate(per_it, start=1):
        kinds = [e[0] for e in log]
        assert kinds[0] == "lr" and log[0][1] == it                                    # train.py:58
        loss = [c for c in cl if c[0] == "loss"][0]
        assert loss[3] == (opt.lambda_normal if it > opt.normal_from_iter else 0.0)    # train.py:77-78 (7000 / 3000 in the reference)
        assert loss[4] == (opt.lambda_dist if it > opt.dist_from_iter else 0.0)
        assert loss[1] == (it > opt.dist_from_iter)                                    # allmap only enters when a regulariser is on
        densified = it < opt.densify_until_iter and it > opt.densify_from_iter and it % opt.densification_interval == 0
        assert ("stats" in kinds) == (it < opt.densify_until_iter)                     # train.py:126-128
        assert ("densify" in kinds) == densified                                       # train.py:130-132
        if densified:
            a = [e for e in log if e[0] == "densify"][0][1]
            assert a[0] == opt.densify_grad_threshold and a[1] == opt.opacity_cull and a[2] == 3.0
            assert a[3] == (20 if it > opt.opacity_reset_interval else None)           # size_threshold, train.py:131
        assert ("reset" in kinds) == (it < opt.densify_until_iter and it % opt.opacity_reset_interval == 0)   # train.py:134-135
        # re-created parameters carry no gradient in the reference -> no update on densification iterations, none on the last one
        assert ("adam" in kinds) == (it < opt.iterations and not densified)            # train.py:137-139
        # statistics + update as one launch exactly where nothing can come between them: no densification, no opacity reset this iteration
        reset = it < opt.densify_until_iter and it % opt.opacity_reset_interval == 0
        assert ("one_launch" in kinds) == (it < opt.iterations and not densified and not reset)
        if "one_launch" in kinds:
            assert kinds.index("adam") > (kinds.index("stats") if "stats" in kinds else -1)
    # one view per iteration, sampled without replacement per epoch (train.py:64-67)
    views = [c[1] for _, cl in per_it for c in cl if c[0] == "raster"]
    for e in range(0, 60, 4):
        assert sorted(views[e:e + 4]) == [0, 1, 2, 3]
    assert m.active_sh_degree == 0                                                      # no multiple of 1000 reached (train.py:61-62)


def test_lazily_counted_iterations_redo_an_overflowing_frame(trainer):
    """Trainer.lazy_count (default): every iteration renders once with OPT_LAZY_COUNT and collects the count after the backward; an
    iteration whose frame is reported as overflowed renders the same view again with exact binning and repeats loss and backward —
    before the statistics and the optimiser step, which still happen once."""
    import surfel_native as n
    tr, m, calls, opt = trainer
    assert tr.lazy_count
    tr._fake_finish_count.overflow_at = (3,)
    for it in range(1, 6):
        n0, c0 = len(m.log), len(calls)
        tr.step()
        rast = [c for c in calls[c0:] if c[0] == "raster"]
        loss = [c for c in calls[c0:] if c[0] == "loss"]
        kinds = [e[0] for e in m.log[n0:]]
        if it == 3:
            assert [c[2] for c in rast] == [n.OPT_LAZY_COUNT, n.OPT_EXACT_BINNING] and rast[0][1] == rast[1][1] and len(loss) == 2
        else:
            assert [c[2] for c in rast] == [n.OPT_LAZY_COUNT] and len(loss) == 1
        assert kinds.count("stats") == 1 and kinds.count("adam") == 1
    assert tr.lazy_overflows == 1
    tr.lazy_count = False
    c0 = len(calls)
    tr.step()
    assert [c[2] for c in calls[c0:] if c[0] == "raster"] == [0]


def test_white_background_resets_opacity_at_densify_start(monkeypatch):
    import surfel_trainer as TR
    monkeypatch.setattr(TR, "rasterize", lambda cam, m, pipe, bg, zero_means2D=True, debug_bits=0: (torch.zeros(3, 4, 4, requires_grad=True),
                                                                                      torch.ones(m.P, dtype=torch.int32), None, torch.zeros(m.P, 3, requires_grad=True)))
    monkeypatch.setattr(TR, "train_loss", lambda image, *a, **k: (image.sum() * 0.0, torch.zeros(6)))
    cams = [types.SimpleNamespace(uid=0, original_image=torch.zeros(3, 4, 4), camera_center=torch.zeros(3), post_consts=lambda: None)]
    m = FakeModel()
    opt = TR.optimization_params(iterations=12, densify_from_iter=10, densification_interval=5, opacity_reset_interval=1000, dist_from_iter=10 ** 6,
                                 normal_from_iter=10 ** 6)
    tr = TR.Trainer(m, cams, opt, white_background=True, extent=1.0)
    tr.manual_chain = False
    assert tr.background.tolist() == [1.0, 1.0, 1.0]
    for _ in range(11):
        tr.step()
    resets = [i for i, e in enumerate(m.log) if e[0] == "reset"]
    assert len(resets) == 1                                                             # at iteration == densify_from_iter (train.py:134)
    # that iteration still takes its optimiser step, with the opacity gradient dropped (the reference re-creates only that parameter)
    assert m.log[resets[0] + 1][0] == "adam" and float(m._gv["opacity"].abs().sum()) == 0.0


def test_early_gather_is_a_measured_choice(trainer):
    """N > 1 start-up probe (VERDICT r2 #8b): "auto" runs EG_LEN iterations with the early colour gather, EG_LEN without, and keeps
    the early form only if it wins by more than noise; a single process never probes."""
    import surfel_trainer as TR
    tr, m, calls, opt = trainer
    assert tr.early_gather is False and tr._probe_early_gather() is False          # world 1: nothing to gather
    assert TR.Trainer.decide_early_gather(0.95, 1.00) and not TR.Trainer.decide_early_gather(0.99, 1.00)
    assert not TR.Trainer.decide_early_gather(1.10, 1.00)
    # the probe's schedule (events and the collective are device-side: checked on the GPU box by the RCCL rehearsal)
    tr.early_gather, tr.iteration, tr._eg_first = "auto", 10, 12
    assert tr._probe_early_gather() is False                                       # still warming up
    # the probe windows avoid densification / opacity-reset iterations: with an event every 5 iterations no window of
    # 2 x EG_LEN + 1 iterations is free while the statistics are live, so the probe starts behind the last event (35; densify_until_iter = 40)
    tr.early_gather, tr._eg_first, tr._eg_events = "auto", None, []
    tr.opt.densify_from_iter, tr.opt.densification_interval, tr.opt.densify_until_iter, tr.opt.opacity_reset_interval = 0, 5, 40, 3000
    tr.iteration = 10
    assert tr._is_event_iteration(15) and not tr._is_event_iteration(16) and not tr._is_event_iteration(45)
    assert tr._probe_early_gather() is False and tr._eg_first == 36
    tr.opt.densification_interval = 100
    tr._eg_first = None
    assert tr._probe_early_gather() is False and tr._eg_first == 12
    # a rank whose early gather failed keeps the probe's state machine running (the verdict's all-reduce needs every rank) but takes
    # the late form — the same collectives, issued behind the backward
    tr._eg_failed = False
    assert tr._eg_failed is False


def test_early_gather_probe_keeps_every_rank_in_the_verdict(trainer, monkeypatch):
    """a rank whose early colour gather failed during the start-up probe must keep the probe's state machine running —
    the other ranks wait for it in the verdict's all-reduce — take the late form from then on, and vote "early = never", so that MAX
    over the ranks turns every rank to the late form.  Events and the collective are faked (no GPU here)."""
    import surfel_trainer as TR
    tr, m, calls, opt = trainer
    reduced = []

    class FakeEvent:
        t = 0.0

        def __init__(self, enable_timing=False): self.stamp = None
        def record(self): FakeEvent.t += 1.0; self.stamp = FakeEvent.t
        def synchronize(self): pass
        def elapsed_time(self, other): return other.stamp - self.stamp
    monkeypatch.setattr(TR.torch.cuda, "Event", FakeEvent)
    monkeypatch.setattr(TR.dist, "all_reduce", lambda t, op=None: reduced.append(t.clone()))
    tr.early_gather, tr._eg_first, tr._eg_events, tr._eg_failed = "auto", None, [], False
    tr.opt.densify_from_iter = 10 ** 9
    took = []
    for it in range(1, 3 + 2 * tr.EG_LEN + 2):
        tr.iteration = it
        if it == 5:      # the hook failed on this rank in the middle of the early window
            tr._eg_failed = True
        took.append(tr._probe_early_gather())
    # warm-up (2 iterations) | early window: early until the failure, late after it | late window | verdict reached exactly once
    assert took[:2] == [False, False] and took[2:4] == [True, True] and took[4:6] == [False, False]
    assert len(reduced) == 1 and float(reduced[0][0]) > 1e37, reduced      # this rank's vote: the early form never wins
    assert tr.early_gather is False and tr.early_gather_probe["choice"] == "late"

