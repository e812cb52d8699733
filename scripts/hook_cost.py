"""Cost of splitting dL/dcolour off the backward (surfel_set_backward_hook) on one GPU: the preprocess_bwd stage (HIP events on the
launch stream) with and without a no-op hook.   python scripts/hook_cost.py [workload ...]"""
import json
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "2d-gaussian-splatting_amd"), os.path.join(REPO, "tests")):
    sys.path.insert(0, p)
import surfel_native as n  # noqa: E402
import synthetic  # noqa: E402
from helpers import HipRun, scene_args  # noqa: E402

for name in (sys.argv[1:] or ["C2", "C4"]):
    P, W, H, zf = synthetic.CONFIGS[name]
    sc = synthetic.make_scene(P, W, H, seed=0, z_far=zf)
    a = scene_args(sc)
    rng = np.random.default_rng(0)
    gC = rng.normal(size=(3, H, W)).astype(np.float32); gO = rng.normal(size=(7, H, W)).astype(np.float32)
    run = HipRun(a).forward()
    run.debug = 2
    row = {"workload": name, "P": P, "R": run.R}
    for label, hook in (("single_kernel_us", None), ("split_us", lambda: None), ("single_kernel_again_us", None)):
        n.set_backward_hook(hook)
        for _ in range(3):
            run.backward(gC, gO)
        n.collect_stage_times()
        for _ in range(10):
            run.backward(gC, gO)
        t = n.collect_stage_times()
        row[label] = round(1e3 * t["preprocess_bwd"][0] / t["preprocess_bwd"][1], 1)
    n.set_backward_hook(None)
    print(json.dumps(row))
