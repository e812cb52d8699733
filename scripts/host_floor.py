"""Host-side cost of one training iteration: the step on a frame so small that the GPU work vanishes (every kernel ~ its launch
latency), plus a cProfile of the same loop.  What this prints is the floor the Python / ctypes / autograd launch path puts under the
C2 step (profiles/r03_host_floor.md).  Usage: python scripts/host_floor.py [steps]"""
import cProfile
import io
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "2d-gaussian-splatting_amd"))
import surfel_trainer as TR  # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    d = torch.device("cuda:0")
    bg = torch.zeros(3, device=d)
    gt = TR.synthetic_object(2000, d, seed=1, px_scale=0.06)
    cams = TR.capture_views(gt, TR.orbit_cameras(8, 64, 64, device=d), bg)
    m = TR.synthetic_object(2000, d, seed=2, px_scale=0.05)
    m.spatial_lr_scale = 1.0
    tr = TR.Trainer(m, cams, TR.optimization_params(dist_from_iter=0, normal_from_iter=0, lambda_dist=10.0, densify_from_iter=10 ** 9,
                                                    densify_until_iter=10 ** 9), TR.pipeline_params(depth_ratio=1.0))
    for _ in range(50):
        tr.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        tr.step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    print("tiny frame (2 000 surfels, 64x64): %.1f us per step (host + launch latencies)" % (dt * 1e6))
    if os.environ.get("HOST_FLOOR_PROFILE", "1") == "0":
        return
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(steps):
        tr.step()
    torch.cuda.synchronize()
    pr.disable()
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28)
    print(s.getvalue()[:6000])


if __name__ == "__main__":
    main()
