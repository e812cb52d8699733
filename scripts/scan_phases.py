#!/usr/bin/env python
"""Phase split of the scan walk on a trainer's frames (diagnostic build: python build.py --variant scantime -DSCAN_TIMING).
    SURFEL_LIB=.../libsurfel_hip_scantime.so SURFEL_OPTIONS=bwd_variant=3,bwd_tune=0 python scripts/scan_phases.py trained garden"""
import json, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "2d-gaussian-splatting_amd")); sys.path.insert(0, REPO)
import torch
import surfel_native as sn
from helpers_bench import TRAINED_PRESETS, make_trainer, trained_trainer
dev = torch.device("cuda:0"); lib = sn.load()
for wl in sys.argv[1:] or ["trained"]:
    tr = trained_trainer(dev, wl)[0] if wl in TRAINED_PRESETS else make_trainer(dev, wl, n_views=8)
    for _ in range(12):
        tr.step()
    st = torch.zeros(8, dtype=torch.int64, device=dev)
    lib.surfel_debug_set_blend_stats(sn.ptr(st))
    for _ in range(4):
        tr.step()
    torch.cuda.synchronize()
    lib.surfel_debug_set_blend_stats(None)
    s = st.cpu().numpy().astype(float)
    names = ["stores of the previous records + next loads issued + ballots", "walk", "barrier behind walk", "flush", "other barriers + ranks/lists", "top barrier + wait for the prefetched records + LDS writes + footprint test"]
    tot = s[6]
    print(json.dumps({"workload": wl, "waves": int(s[7]), "cycles_per_wave": round(tot / max(1, s[7])), **{n: round(s[i] / tot, 3) for i, n in enumerate(names)}}), flush=True)
    del tr
    import diff_surfel_rasterization as dsr
    dsr.set_grad_arena(None); torch.cuda.empty_cache()
