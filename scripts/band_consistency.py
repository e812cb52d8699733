"""Tile-band sharding rehearsal (BASELINE config 5 path): N ranks render row bands of the SAME view, the gathered image feeds the loss,
every rank back-propagates its rows; the exchanged per-surfel gradient must equal the single-GPU gradient of the full view, and a
short band-sharded training must behave like training.

    SURFEL_DIST_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 scripts/band_consistency.py
"""
import json, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "2d-gaussian-splatting_amd")); sys.path.insert(0, REPO)
import numpy as np, torch, torch.distributed as dist
import surfel_dist, surfel_model, surfel_trainer as TR
from surfel_losses import train_loss
from surfel_render import rasterize

rank, world, local = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("LOCAL_RANK", 0))
backend = os.environ.get("SURFEL_DIST_BACKEND", "nccl")
local = local % torch.cuda.device_count() if backend != "nccl" else local
torch.cuda.set_device(local); dev = torch.device("cuda", local)
dist.init_process_group(backend)
bg = torch.zeros(3, device=dev)
gt = TR.synthetic_object(4000, dev, seed=0, px_scale=0.06)
cams = TR.capture_views(gt, TR.orbit_cameras(6, 128, 112, device=dev), bg)
g = torch.Generator().manual_seed(5)


def fresh():
    m = surfel_model.GaussianModel(3, device=dev)
    m.set_parameters(gt._xyz.cpu() + 0.02 * torch.randn((gt.P, 3), generator=torch.Generator().manual_seed(5)), gt._features_dc.cpu() * 0.5,
                     gt._features_rest.cpu() * 0.0, gt._opacity.cpu() - 1.0, gt._scaling.cpu(), gt._rotation.cpu())
    m.active_sh_degree = 3; m.spatial_lr_scale = 1.0
    m.training_setup(TR.optimization_params())
    return m


cam = cams[1]; pipe = TR.pipeline_params(depth_ratio=1.0)
# (1) band path by hand: band render -> gather -> full-image loss -> backward -> exchange -> explicit SH gradients
m = fresh(); m.bind(sh_grad=False)
bounds = surfel_dist.band_bounds(int(cam.image_height), world)
img, radii, am, m2 = rasterize(cam, m, pipe, bg, band=bounds[rank])
loss, sc = train_loss(surfel_dist.gather_bands(img, bounds), surfel_dist.gather_bands(am, bounds), cam.original_image, cam.post_consts(), 1.0, 0.2, 0.05, 100.0)
loss.backward()
gall = surfel_model.exchange_collectives(m.grad, m.gcol, m.P)
m.sh_grad_from_colours(cam.camera_center[None].expand(world, 3).contiguous(), gall)
g_band = m.grad.clone()
# (2) the same view unsharded on this rank
f = fresh(); f.bind(sh_grad=True)
img, radii, am, m2 = rasterize(cam, f, pipe, bg)
loss_f, sc_f = train_loss(img, am, cam.original_image, cam.post_consts(), 1.0, 0.2, 0.05, 100.0)
loss_f.backward()
g_full = f.grad
cos = float(torch.nn.functional.cosine_similarity(g_band, g_full, dim=0))
scale = float(g_full.abs().mean())
frac = float(((g_band - g_full).abs() <= 1e-3 * scale + 2e-3 * g_full.abs()).float().mean())
# (3) a short band-sharded training run
tr = TR.Trainer(fresh(), cams, TR.optimization_params(iterations=150, densify_from_iter=50, densification_interval=50, densify_until_iter=130,
                                                      opacity_reset_interval=10 ** 6, dist_from_iter=20, normal_from_iter=40, lambda_dist=10.0),
                pipe, sharding="bands")
p0 = tr.evaluate()[0]
for _ in range(150):
    tr.step()
p1 = tr.evaluate()[0]
digest = float(tr.model.theta.double().sum())
ds = [None] * world
dist.all_gather_object(ds, (digest, tr.model.P))
if rank == 0:
    print(json.dumps({"world": world, "loss_band": float(sc[5]), "loss_full": float(sc_f[5]), "grad_cosine": cos, "grad_frac_close": frac,
                      "psnr_before": round(p0, 3), "psnr_after": round(p1, 3), "replicas_identical": len(set(ds)) == 1, "points": [d[1] for d in ds]}))
dist.destroy_process_group()
