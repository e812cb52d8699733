"""View-parallel consistency rehearsal: N ranks (gloo or nccl) train the same synthetic capture with densification; at the end every
rank's parameter store must be BIT-IDENTICAL (identical exchanged gradients -> identical Adam steps -> identical densification).

    SURFEL_DIST_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 scripts/dist_consistency.py [iters]
"""
import hashlib, json, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "2d-gaussian-splatting_amd")); sys.path.insert(0, REPO)
import numpy as np, torch, torch.distributed as dist
import surfel_model, surfel_trainer as TR

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 350
rank, world, local = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("LOCAL_RANK", 0))
backend = os.environ.get("SURFEL_DIST_BACKEND", "nccl")
local = local % torch.cuda.device_count() if backend != "nccl" else local
torch.cuda.set_device(local); dev = torch.device("cuda", local)
dist.init_process_group(backend)
torch.manual_seed(1234 + rank)            # deliberately different global RNG streams: the trainer must not depend on them
bg = torch.zeros(3, device=dev)
gt = TR.synthetic_object(4000, dev, seed=0, px_scale=0.06)
cams = TR.capture_views(gt, TR.orbit_cameras(12, 128, 112, device=dev), bg)
rng = np.random.default_rng(0)
pcd = type("PCD", (), {})(); pcd.points = (gt._xyz.cpu().numpy() + 0.05 * rng.normal(size=(gt.P, 3))).astype(np.float32)
pcd.colors = np.full((gt.P, 3), 0.5, np.float32)
model = surfel_model.GaussianModel(3, device=dev)
g = torch.Generator(device=dev); g.manual_seed(7)
torch.manual_seed(7)                       # create_from_pcd draws the initial rotations from the global RNG (as the reference does)
model.create_from_pcd(pcd, spatial_lr_scale=TR.cameras_extent(cams))
torch.manual_seed(1234 + rank)
opt = TR.optimization_params(iterations=iters, densify_from_iter=100, densification_interval=50, densify_until_iter=iters - 20,
                             opacity_reset_interval=200, dist_from_iter=60, normal_from_iter=120, lambda_dist=10.0, position_lr_max_steps=iters)
tr = TR.Trainer(model, cams, opt, TR.pipeline_params(depth_ratio=1.0))
p0 = tr.evaluate()[0]
for _ in range(iters):
    tr.step()
torch.cuda.synchronize()
digest = hashlib.sha256(model.theta.cpu().numpy().tobytes() + model.m.cpu().numpy().tobytes() + model.v.cpu().numpy().tobytes()).hexdigest()
digests = [None] * world
dist.all_gather_object(digests, (digest, model.P))
if rank == 0:
    print(json.dumps({"world": world, "backend": backend, "iterations": iters, "points": [d[1] for d in digests], "identical": len(set(digests)) == 1,
                      "psnr_before": round(p0, 3), "psnr_after": round(tr.evaluate()[0], 3), "sha256": digest[:16]}))
dist.destroy_process_group()
