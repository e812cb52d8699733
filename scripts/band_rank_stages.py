#!/usr/bin/env python
"""What ONE rank of the tile-band mode computes (VERDICT r4 #7 / DESIGN section 6): the rasterizer forward + backward of a single row band
of a frame, stage by stage, next to the full frame — on one GPU, no collectives.  The P-sized stages that every rank repeats
(preprocess over all surfels, scans, Adam) against the band-divisible ones (emission, sorts, blend) give the measured Amdahl table.
    python scripts/band_rank_stages.py [workload=C5] [ranks=8]"""
import json
import os
import sys

import torch

REPO = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(REPO, "2d-gaussian-splatting_amd")); sys.path.insert(0, REPO)
import surfel_native as n  # noqa: E402
import surfel_dist  # noqa: E402
from helpers_bench import make_trainer  # noqa: E402
from surfel_render import rasterize  # noqa: E402


def stages(tr, cam, band, reps=6):
    m = tr.model
    tr.pipe.debug = 2
    out = {}
    for rep in range(reps + 2):
        m.bind(sh_grad=False)
        image, radii, allmap, means2D = rasterize(cam, m, tr.pipe, tr.background, zero_means2D=False, band=band)
        gi, ga = torch.randn_like(image), torch.randn_like(allmap)
        torch.autograd.backward([image, allmap], [gi, ga])
        torch.cuda.synchronize()
        st = n.collect_stage_times()
        if rep >= 2:
            for k, v in st.items():
                out.setdefault(k, []).append(v[0] / v[1])
    tr.pipe.debug = 0
    import diff_surfel_rasterization as dsr
    return {k: round(sum(v) / len(v), 4) for k, v in out.items()}, int(dsr.last_num_rendered), int((radii > 0).sum().item())


def main():
    wl = sys.argv[1] if len(sys.argv) > 1 else "C5"
    ranks = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    dev = torch.device("cuda:0")
    tr = make_trainer(dev, wl, n_views=2)
    for _ in range(3):
        tr.step()
    torch.cuda.synchronize()
    cam = tr.cams[0]
    H = int(cam.image_height)
    full, R, V = stages(tr, cam, None)
    print(json.dumps({"workload": wl, "part": "full frame", "instances": R, "visible": V, "stages_ms": full, "sum_ms": round(sum(full.values()), 3)}), flush=True)
    bounds = surfel_dist.band_bounds(H, ranks, None, multiple=surfel_dist.HALO)
    worst = None
    for k in sorted(set([0, ranks // 2, ranks - 1])):
        st, Rb, Vb = stages(tr, cam, bounds[k])
        rec = {"workload": wl, "part": "band %d of %d: rows [%d, %d)" % (k, ranks, bounds[k][0], bounds[k][1]), "instances": Rb, "visible": Vb, "stages_ms": st,
               "sum_ms": round(sum(st.values()), 3)}
        print(json.dumps(rec), flush=True)
        if worst is None or rec["sum_ms"] > worst["sum_ms"]:
            worst = rec
    # the optimiser step every rank repeats on all P surfels
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    m = tr.model
    m.bind(sh_grad=False)
    image, radii, allmap, means2D = rasterize(cam, m, tr.pipe, tr.background, zero_means2D=False)
    torch.autograd.backward([image, allmap], [torch.randn_like(image), torch.randn_like(allmap)])
    with torch.no_grad():
        for _ in range(2):
            m.optimizer_step(grad_scale=1.0, colour_grads=(cam.camera_center[None], m.gcol[None]))
        e0.record()
        for _ in range(5):
            m.optimizer_step(grad_scale=1.0, colour_grads=(cam.camera_center[None], m.gcol[None]))
        e1.record()
    torch.cuda.synchronize()
    adam = e0.elapsed_time(e1) / 5
    rast_full, rast_band = sum(full.values()), worst["sum_ms"]
    print(json.dumps({"workload": wl, "ranks": ranks, "adam_ms_all_surfels": round(adam, 3), "rasterizer_full_ms": round(rast_full, 3),
                      "rasterizer_heaviest_band_ms": round(rast_band, 3),
                      "rank_local_compute_ms (heaviest band + Adam; loss share and collectives not included)": round(rast_band + adam, 3),
                      "single_gpu_compute_ms (full frame + Adam)": round(rast_full + adam, 3),
                      "compute_speedup_bound_at_%d_ranks" % ranks: round((rast_full + adam) / (rast_band + adam), 2)}), flush=True)


if __name__ == "__main__":
    main()
