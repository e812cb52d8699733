#!/usr/bin/env python
"""Per-tile list statistics of a workload's frames: list length n, deepest composited position maxc (= what the blend passes
stage), per-pixel depth of the last contributor — the data behind the heavy-tile split.  GPU only.

    python scripts/tile_lists.py trained garden C2 -> gpurun_out/tile_lists_<workload>.npz + a summary on stdout
"""
import json
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "2d-gaussian-splatting_amd"))
sys.path.insert(0, REPO)


def frame_stats(W, H):
    """(n, maxc, sum of last per tile) of the most recent forward, from the image buffer (white box, as dsr.staged_instances)."""
    import torch
    import diff_surfel_rasterization as dsr
    buf, gx, gy = dsr._last_image
    ranges = buf[:gx * gy * 8].view(torch.int32).view(gy * gx, 2)
    n = (ranges[:, 1] - ranges[:, 0]).cpu().numpy()
    off = dsr.image_layout(W, H)[3]
    last = buf[off:off + 4 * W * H].view(torch.int32).view(H, W)
    pad = torch.zeros((gy * 16, gx * 16), dtype=torch.int32, device=buf.device)
    pad[:H, :W] = last
    t = pad.view(gy, 16, gx, 16).permute(0, 2, 1, 3).reshape(gy * gx, 256)
    return n, t.amax(dim=1).cpu().numpy(), t.to(torch.int64).sum(dim=1).cpu().numpy()


def main():
    import torch
    import synthetic
    import surfel_native
    from helpers_bench import TRAINED_PRESETS, make_trainer, trained_trainer
    dev = torch.device("cuda:0")
    os.makedirs(os.path.join(REPO, "gpurun_out"), exist_ok=True)
    for wl in sys.argv[1:] or ["trained"]:
        if wl in TRAINED_PRESETS:
            tr, _ = trained_trainer(dev, wl)
            W, H = TRAINED_PRESETS[wl]["res"]
        else:
            P, W, H, zf = synthetic.CONFIGS[wl]
            tr = make_trainer(dev, wl, n_views=8)
        for _ in range(10):
            tr.step()
        tr.pipe.debug = 2
        surfel_native.collect_stage_times()
        frames = []
        for _ in range(8):
            tr.step()
            torch.cuda.synchronize()
            frames.append(frame_stats(W, H))
        st = surfel_native.collect_stage_times()
        tr.pipe.debug = 0
        n = np.stack([f[0] for f in frames]); mc = np.stack([f[1] for f in frames]); sl = np.stack([f[2] for f in frames])
        np.savez_compressed(os.path.join(REPO, "gpurun_out", "tile_lists_%s.npz" % wl), n=n, maxc=mc, sumlast=sl)
        f0n, f0m, f0s = n[0], mc[0], sl[0]
        order = np.argsort(-f0m)
        summ = {"workload": wl, "P": int(tr.model.P), "tiles": int(f0n.size), "R": int(f0n.sum()), "staged": int(f0m.sum()),
                "pairs_upper_bound_Mlast": round(float(f0s.sum()) / 1e6, 2),
                "maxc_top10": f0m[order[:10]].tolist(), "n_of_those": f0n[order[:10]].tolist(),
                "mean_last_over_maxc_top10": [round(float(f0s[i]) / 256.0 / max(1, f0m[i]), 3) for i in order[:10]],
                "maxc_percentiles_50_90_99_max": [int(np.percentile(f0m, q)) for q in (50, 90, 99, 100)],
                "tiles_with_maxc_over": {str(k): int((f0m > k).sum()) for k in (256, 512, 1024, 2048, 4096)},
                "staged_in_tiles_over": {str(k): int(f0m[f0m > k].sum()) for k in (256, 512, 1024, 2048, 4096)},
                "kernel_ms": {k: round(v[0] / v[1], 4) for k, v in st.items()}}
        print(json.dumps(summ), flush=True)
        del tr
        import diff_surfel_rasterization as dsr
        dsr.set_grad_arena(None)
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
