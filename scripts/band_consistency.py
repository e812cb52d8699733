"""Tile-band sharding rehearsal (BASELINE config 5 path): N ranks render row bands of the SAME view, each evaluates the loss on its band
(+ 32-row halo from its neighbours) and back-propagates its rows; the per-surfel gradient after ONE 52 B/surfel all-reduce must equal
the single-GPU gradient of the full view, and a short band-sharded training must behave like training.

    SURFEL_DIST_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 scripts/band_consistency.py
"""
import json, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "2d-gaussian-splatting_amd")); sys.path.insert(0, REPO)
import numpy as np, torch, torch.distributed as dist
import surfel_dist, surfel_model, surfel_trainer as TR
from surfel_losses import scalars_from_band_sums, train_loss, train_loss_band
from surfel_render import post_consts_rows, rasterize

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
# (1) band path by hand: band render -> halo exchange -> band share of the loss -> backward -> ONE all-reduce of 52 B/surfel ->
# explicit SH gradients from the summed colour gradients
m = fresh(); m.bind(sh_grad=False)
H, W = int(cam.image_height), int(cam.image_width)
bounds = surfel_dist.band_bounds(H, world, multiple=surfel_dist.HALO)
y0, y1 = bounds[rank]
img, radii, am, m2 = rasterize(cam, m, pipe, bg, band=(y0, y1))
ext = surfel_dist.exchange_halo(torch.cat([img, am], 0), bounds, H)
top, bot = surfel_dist.halo_rows(bounds, rank, H)
share, sums = train_loss_band(ext[:3], ext[3:], cam.original_image[:, y0 - top:y1 + bot], post_consts_rows(cam.post_consts(), y0 - top), 1.0, 0.2, 0.05,
                              100.0, (top, top + y1 - y0), (H, W))
share.backward()
sums = sums.clone(); dist.all_reduce(sums)
sc = scalars_from_band_sums(sums, float(3 * H * W), float(H * W), 0.2, 0.05, 100.0)
surfel_model.exchange_same_view(m.grad, m.P)
m.sh_grad_from_colours(cam.camera_center[None], m.gcol[None])
g_band = m.grad.clone()
# (2) the same view unsharded on this rank
f = fresh(); f.bind(sh_grad=True)
img, radii, am, m2 = rasterize(cam, f, pipe, bg)
loss_f, sc_f = train_loss(img, am, cam.original_image, cam.post_consts(), 1.0, 0.2, 0.05, 100.0)
loss_f.backward()
g_full = f.grad
# compare the parameter gradients (the band path keeps exchange scratch — colour / means2D blocks — in the SH section's head;
# sh_grad_from_colours has just overwritten it with the SH gradients, so the stores are comparable as a whole)
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
                      "psnr_before": round(p0, 3), "psnr_after": round(p1, 3), "replicas_identical": len(set(ds)) == 1, "points": [d[1] for d in ds],
                      "wire_bytes_last_step": tr.wire, "band_bounds_last": surfel_dist.band_bounds(H, world, tr._row_weights, multiple=surfel_dist.HALO)}))
dist.destroy_process_group()
