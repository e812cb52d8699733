import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, "2d-gaussian-splatting_amd")); sys.path.insert(0, REPO)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29543")
import torch, torch.distributed as dist
import surfel_trainer as TR
d = torch.device("cuda", 0); torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=d)
bg = torch.zeros(3, device=d)
gt_model = TR.synthetic_object(4000, d, seed=1, px_scale=0.06)
cams = TR.capture_views(gt_model, TR.orbit_cameras(4, 128, 96, device=d), bg)
hist = []
for early in (False, False, True, True):
    m = TR.synthetic_object(4000, d, seed=2, px_scale=0.05)
    m.spatial_lr_scale = 1.0
    tr = TR.Trainer(m, cams, TR.optimization_params(dist_from_iter=0, normal_from_iter=0, lambda_dist=10.0, densify_from_iter=10 ** 9),
                    TR.pipeline_params(depth_ratio=1.0), rehearse_exchange=True)
    tr.early_gather = early
    snaps = []
    for _ in range(4):
        tr.step(); torch.cuda.synchronize()
        snaps.append((m.theta.clone(), m.grad.clone(), tr.last["scalars"].clone()))
    hist.append(snaps)
for j in range(1, 4):
    for it in range(4):
        a, b = hist[0][it], hist[j][it]
        print("run", j, "it", it, "theta diff", float((a[0] - b[0]).abs().max()), "nonequal", int((a[0] != b[0]).sum()), "loss", a[2][5].item(), b[2][5].item())
dist.destroy_process_group()
