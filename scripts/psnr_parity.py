"""PSNR-parity experiment: the reference trains on 1 view per iteration
(/root/reference/train.py:64-69); view-parallel training consumes N views per step.  Same scene, same seed:

    rule "steps"  : averaged gradients, the reference's schedule unchanged (K steps -> N x more images seen)
    rule "images" : averaged gradients, schedule divided by N (surfel_trainer.scale_schedule: K/N steps, same images seen)

    rule "images-lrN" / "images-lrsqrt" : as "images" with every learning rate x N / x sqrt(N)

    python scripts/psnr_parity.py RULE [K] [N]         N virtual ranks on one process (Trainer.views_per_step: the same averaged
                                                       N-view step, no collectives; N = 1 is the reference's loop)
    SURFEL_DIST_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 scripts/psnr_parity.py RULE [K]

One JSON line per evaluation point on rank 0: train PSNR (8 views), held-out PSNR (8 views never trained on), points."""
import json, os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "2d-gaussian-splatting_amd")); sys.path.insert(0, REPO)
import numpy as np, torch, torch.distributed as dist
import surfel_model, surfel_trainer as TR

rule = sys.argv[1] if len(sys.argv) > 1 else "steps"
K = int(sys.argv[2]) if len(sys.argv) > 2 else 6000
virtual = int(sys.argv[3]) if len(sys.argv) > 3 else 1
n_gt, n_views, res = 30000, 40, 400
rank, world, local = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("LOCAL_RANK", 0))
backend = os.environ.get("SURFEL_DIST_BACKEND", "nccl")
local = local % torch.cuda.device_count() if backend != "nccl" else local
torch.cuda.set_device(local); dev = torch.device("cuda", local)
if world > 1:
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group(backend)
torch.manual_seed(0)
bg = torch.zeros(3, device=dev)
gt = TR.synthetic_object(n_gt, dev, seed=0, px_scale=0.035)
cams = TR.capture_views(gt, TR.orbit_cameras(n_views + 8, res, res, device=dev), bg)
del gt
test_cams = cams[::6]                      # every 6th view of the orbit is held out (interpolation, not a missing sector)
train_cams = [c for i, c in enumerate(cams) if i % 6 != 0]
extent = TR.cameras_extent(train_cams)
rng = np.random.default_rng(0)
pcd = type("PCD", (), {})()
pcd.points = (rng.random((n_gt, 3)) * 2.6 - 1.3).astype(np.float32)
pcd.colors = rng.random((n_gt, 3)).astype(np.float32)
model = surfel_model.GaussianModel(3, device=dev)
model.create_from_pcd(pcd, spatial_lr_scale=extent)          # torch.manual_seed(0) above: identical random rotations on every rank
opt = TR.optimization_params(iterations=K, lambda_dist=100.0, position_lr_max_steps=K, densify_until_iter=K // 2,
                             opacity_reset_interval=K // 2, dist_from_iter=K // 10, normal_from_iter=K * 7 // 30)
N = world * virtual
if rule.startswith("images"):
    opt = TR.scale_schedule(opt, N, lr="linear" if rule.endswith("lrN") else ("sqrt" if rule.endswith("lrsqrt") else "none"))
tr = TR.Trainer(model, train_cams, opt, TR.pipeline_params(depth_ratio=1.0), extent=extent)
tr.views_per_step = virtual
steps = opt.iterations
every = max(1, steps // 12)
t0 = time.perf_counter()
for it in range(1, steps + 1):
    tr.step()
    if it % every == 0 or it == steps:
        pt, ph = tr.evaluate(train_cams[:8])[0], tr.evaluate(test_cams)[0]
        if rank == 0:
            print(json.dumps({"rule": rule, "world": N, "processes": world, "K": K, "step": it, "steps": steps, "images_seen": it * N, "points": model.P,
                              "psnr_train": round(pt, 3), "psnr_heldout": round(ph, 3), "wall_s": round(time.perf_counter() - t0, 1)}), flush=True)
if world > 1:
    digest = float(model.theta.double().sum()); ds = [None] * world
    dist.all_gather_object(ds, (digest, model.P))
    if rank == 0:
        print(json.dumps({"rule": rule, "world": world, "replicas_identical": len(set(ds)) == 1}), flush=True)
    dist.destroy_process_group()
