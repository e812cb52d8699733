"""preprocess_bwd: SH block by per-thread loads vs LDS-DMA (run ON THE GPU BOX).  python scripts/pbwd_dma_ab.py [workload ...]"""
import json, os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "2d-gaussian-splatting_amd"), os.path.join(REPO, "tests")):
    sys.path.insert(0, p)
import torch
import surfel_native as n
import synthetic
from helpers import HipRun, scene_args

for name in (sys.argv[1:] or ["C4"]):
    P, W, H, zf = synthetic.CONFIGS[name]
    a = scene_args(synthetic.make_scene(P, W, H, seed=0, z_far=zf, px_radius=synthetic.PX_RADIUS.get(name)))
    rng = np.random.default_rng(0)
    gC = rng.normal(size=(3, H, W)).astype(np.float32); gO = rng.normal(size=(7, H, W)).astype(np.float32)
    row = {"workload": name, "P": P}
    for rep in range(2):
        for flag, nm in ((n.OPT_PBWD_NO_DMA, "loads"), (n.OPT_PBWD_DMA, "dma")):
            run = HipRun(a, debug=2 | flag)
            run.forward(); run.backward(gC, gO)
            n.collect_stage_times()
            for _ in range(5):
                run.forward(); run.backward(gC, gO)
            t = n.collect_stage_times()
            row["%s_%d" % (nm, rep)] = {k: round(1e3 * v[0] / v[1], 1) for k, v in t.items() if k in ("preprocess_bwd", "preprocess_fwd", "blend_bwd")}
            del run
            torch.cuda.empty_cache()
    print(json.dumps(row), flush=True)
