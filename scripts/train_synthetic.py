"""End-to-end training run on a synthetic capture with the reference's default schedule (arguments/__init__.py:75-95):
random-point initialisation as scene/dataset_readers.py:236-242 does for Blender scenes, densification from iteration 500 every
100, opacity reset every 3000, lambda_dist after 3000 / lambda_normal after 7000.  Prints one JSON line per evaluation.

    python scripts/train_synthetic.py [iterations] [gt_surfels] [views] [res]
"""
import json, os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "2d-gaussian-splatting_amd")); sys.path.insert(0, REPO)
import numpy as np, torch
import surfel_model, surfel_trainer as TR

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
n_gt = int(sys.argv[2]) if len(sys.argv) > 2 else 30000
n_views = int(sys.argv[3]) if len(sys.argv) > 3 else 40
res = int(sys.argv[4]) if len(sys.argv) > 4 else 400
torch.manual_seed(0)
dev = torch.device("cuda:0")
bg = torch.zeros(3, device=dev)
gt = TR.synthetic_object(n_gt, dev, seed=0, px_scale=0.035)
cams = TR.capture_views(gt, TR.orbit_cameras(n_views + 8, res, res, device=dev), bg)
train_cams, test_cams = cams[:n_views], cams[n_views:]
extent = TR.cameras_extent(train_cams)
rng = np.random.default_rng(0)
pcd = type("PCD", (), {})()
pcd.points = (rng.random((n_gt, 3)) * 2.6 - 1.3).astype(np.float32)        # random points in the bounding cube, random colours
pcd.colors = rng.random((n_gt, 3)).astype(np.float32)
model = surfel_model.GaussianModel(3, device=dev)
model.create_from_pcd(pcd, spatial_lr_scale=extent)
opt = TR.optimization_params(iterations=iters, lambda_dist=100.0, position_lr_max_steps=iters)
tr = TR.Trainer(model, train_cams, opt, TR.pipeline_params(depth_ratio=1.0), extent=extent)
print(json.dumps({"event": "start", "gt_surfels": n_gt, "views": n_views, "res": res, "init_points": model.P, "extent": round(extent, 3),
                  "psnr_train": round(tr.evaluate(train_cams[:8])[0], 3), "psnr_test": round(tr.evaluate(test_cams)[0], 3)}), flush=True)
torch.cuda.synchronize(); t0 = time.perf_counter(); last_t, last_it = t0, 0
for it in range(1, iters + 1):
    tr.step()
    if it % 500 == 0 or it == iters:
        torch.cuda.synchronize(); now = time.perf_counter()
        sc = tr.last["scalars"].cpu().numpy()
        print(json.dumps({"event": "eval", "iteration": it, "points": model.P, "it_per_s": round((it - last_it) / (now - last_t), 1),
                          "Ll1": round(float(sc[0]), 5), "ssim": round(float(sc[1]), 4), "normal_err": round(float(sc[2]), 4),
                          "dist": round(float(sc[3]), 6), "psnr_train": round(tr.evaluate(train_cams[:8])[0], 3),
                          "psnr_test": round(tr.evaluate(test_cams)[0], 3)}), flush=True)
        torch.cuda.synchronize(); last_t, last_it = time.perf_counter(), it
print(json.dumps({"event": "done", "iterations": iters, "wall_s": round(time.perf_counter() - t0, 2), "points": model.P}), flush=True)
