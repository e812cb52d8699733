"""Host-side cost of the large-sort implementations: wall time of a synchronised rasterizer forward at C4 with the library's own
passes vs rocPRIM (HIP-event stage times do not see host time spent inside the sort call while the device idles)."""
import os, sys, time, json
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "2d-gaussian-splatting_amd"), os.path.join(REPO, "tests")):
    sys.path.insert(0, p)
import torch
import surfel_native as n
import synthetic
from helpers import HipRun, scene_args
lib = n.load()
for name in (sys.argv[1:] or ["C4"]):
    P, W, H, zf = synthetic.CONFIGS[name]
    a = scene_args(synthetic.make_scene(P, W, H, seed=0, z_far=zf))
    for impl in (0, 1, 2, 0, 1):
        lib.surfel_set_option(b"large_sort", impl)
        run = HipRun(a)
        for _ in range(3):
            run.forward()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10):
            run.forward()          # forward() synchronises
        dt = (time.perf_counter() - t0) / 10
        print(json.dumps({"workload": name, "large_sort": impl, "forward_wall_ms": round(dt * 1e3, 3)}), flush=True)
        del run
lib.surfel_set_option(b"large_sort", 2)
