"""RCCL plumbing check on ONE GPU (world_size 1, backend nccl = RCCL): the exact collective calls of the N > 1 training step —
asynchronous all_gather_into_tensor / all_reduce on views of the flat gradient store with stream-level waits, the replica sync,
the statistics reductions — run against the real backend, so that the first multi-GPU launch does not trip over API misuse.
(Point-to-point halo exchange needs >= 2 ranks and is covered by the gloo tests only.)   python scripts/rccl_selfcheck.py"""
import json, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "2d-gaussian-splatting_amd")); sys.path.insert(0, REPO)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
import torch, torch.distributed as dist
import surfel_model, surfel_trainer as TR

dev = torch.device("cuda", 0); torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
bg = torch.zeros(3, device=dev)
gt = TR.synthetic_object(5000, dev, seed=0, px_scale=0.06)
cams = TR.capture_views(gt, TR.orbit_cameras(4, 128, 96, device=dev), bg)
gt.spatial_lr_scale = 1.0
tr = TR.Trainer(gt, cams, TR.optimization_params(dist_from_iter=0, normal_from_iter=0, lambda_dist=10.0), TR.pipeline_params(depth_ratio=1.0))
for _ in range(3):
    tr.step()
m = tr.model
ok = {}
g0 = m.grad.clone(); c0 = m.gcol.clone()
gall, wg, wr = surfel_model.exchange_collectives(m.grad, m.gcol, m.P, async_op=True)
tr.time_exchange = True
tr._timed_wait(wg); tr._timed_wait(wr)
torch.cuda.synchronize()
ok["all_gather_colour"] = bool(torch.equal(gall[0], c0))
ok["all_reduce_geometry"] = bool(torch.equal(m.grad, g0))
ok["exposed_ms"] = round(sum(a.elapsed_time(b) for a, b in tr.exchange_events), 4)
w = surfel_model.exchange_same_view(m.grad, m.P, async_op=True); w.wait(); torch.cuda.synchronize()
ok["all_reduce_52B"] = bool(torch.equal(m.grad, g0))
tr._sync_replicas(); ok["sync_replicas"] = True
tr._reduce_stats(); ok["reduce_stats"] = True
r = torch.arange(5, dtype=torch.int32, device=dev); dist.all_reduce(r, op=dist.ReduceOp.MAX); ok["max_int32"] = bool(r[4] == 4)
dist.barrier(); torch.cuda.synchronize()
tr.step(); ok["step_after"] = bool(torch.isfinite(tr.last["scalars"]).all())
# the whole view-parallel step against RCCL (world 1): early all-gather from inside the backward, split Adam
m2 = TR.synthetic_object(5000, dev, seed=3, px_scale=0.05); m2.spatial_lr_scale = 1.0
tr2 = TR.Trainer(m2, cams, TR.optimization_params(dist_from_iter=0, normal_from_iter=0, lambda_dist=10.0), TR.pipeline_params(depth_ratio=1.0),
                 rehearse_exchange=True)
tr2.time_exchange = True
for _ in range(5):
    tr2.step()
torch.cuda.synchronize()
ok["rehearsed_steps_finite"] = bool(torch.isfinite(tr2.last["scalars"]).all() and torch.isfinite(m2.theta).all())
ok["rehearsed_exposed_ms_per_step"] = round(sum(a.elapsed_time(b) for a, b in tr2.exchange_events) / 5, 4)
print(json.dumps(ok))
dist.destroy_process_group()
